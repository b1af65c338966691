"""Register / LDS / scratch usage per kernel from the device assembly (hipcc --save-temps)."""
import re, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
    name, body = m.group(1), m.group(2)
    if pat and not re.search(pat, name):
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    print(f"{name[:70]:70s} vgpr_total={g('next_free_vgpr'):>4s} accum_off={g('accum_offset'):>4s} sgpr={g('next_free_sgpr'):>4s} "
          f"scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size')}")
