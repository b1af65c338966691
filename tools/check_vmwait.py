#!/usr/bin/env python3
"""check_vmwait.py: the hidden-prefetch kernels (proj_p.hpp k_enc_p / k_dec_p: loads the compiler cannot see + s_waitcnt vmcnt(N)
by hand, scan_quad.hpp vm_wait) are only right while the compiler issues exactly the store instructions the counts were
taken from: N stores per wave on a full tile, the same number again on the guarded path of a partial tile.  If a later
compiler merged or split those stores, vm_wait<N> would return before the prefetch has landed.  This compiles the device
code to assembly (no GPU needed) and checks, per kernel, that the hand-written wait is there and that the kernel holds
2 x N tile stores (+ the few prologue stores).  Exit code 0 = consistent."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sparsernns_amd", "csrc", "s5fxp_api.hip")
# mangled prefix -> (stores per wave and full tile = the wait count, prologue stores allowed on top of 2 x that)
EXPECT = {
    "_ZN2s57k_dec_pILi3ELb0EE": (48, 0), "_ZN2s57k_dec_pILi6ELb0EE": (48, 0),
    "_ZN2s57k_dec_pILi3ELb1EE": (48, 2), "_ZN2s57k_dec_pILi6ELb1EE": (48, 2),
    "_ZN2s57k_enc_pILi3EE": (4, 0), "_ZN2s57k_enc_pILi6EE": (8, 0),
}


def main() -> int:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S",
                               "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = 0
    for key, (n, extra) in EXPECT.items():
        m = re.search(r"^" + re.escape(key) + r"[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
        if not m:
            print(f"{key}: kernel not found")
            bad += 1
            continue
        body = m.group(1)
        stores = len(re.findall(r"^\s*global_store", body, re.M))
        waits = len(re.findall(rf"s_waitcnt vmcnt\({n}\)\s*$", body, re.M))
        ok = waits >= 1 and stores == 2 * n + extra
        print(f"{key}: {stores} stores (expected {2 * n + extra}), {waits} x vmcnt({n}): {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
