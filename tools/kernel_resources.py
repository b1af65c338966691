#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the library's one translation unit (hipcc -Rpass-analysis).

usage: tools/kernel_resources.py [substring ...]   (no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sparsernns_amd", "csrc", "s5fxp_api.hip")


def main() -> None:
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    pats = [a for a in sys.argv[1:] if not a.startswith("-D")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", SRC] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            rows[cur] = {}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
    for name, r in rows.items():
        short = re.sub(r"\(.*", "", name).replace("void s5::", "")
        if pats and not any(p in short for p in pats):
            continue
        print(f"{short:70s} vgpr {r.get('vgpr', -1):3d} agpr {r.get('agpr', 0):3d} sgpr {r.get('sgpr', -1):3d} "
              f"scratch {r.get('scratch', 0):4d} occ {r.get('occ', -1)} lds {r.get('lds', 0)}")


if __name__ == "__main__":
    main()
