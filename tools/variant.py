#!/usr/bin/env python3
"""variant.py NAME [-D...] [FILE OLD NEW]...: builds tools/bin/NAME/libs5fxp.so from a copy of csrc/ in which every (FILE,
OLD, NEW) triple has been applied as an exact string replacement (it must match exactly once).  For A/B timing and
ablations of the shipped kernels (tools/run_variants.sh, tools/ab_bench.sh); results of an ablated build are wrong by design."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    name, rest = sys.argv[1], sys.argv[2:]
    flags = [a for a in rest if a.startswith("-D")]
    rest = [a for a in rest if not a.startswith("-D")]
    assert len(rest) % 3 == 0, "FILE OLD NEW triples"
    w = f"/tmp/v/{name}"
    shutil.rmtree(w, ignore_errors=True)
    os.makedirs(f"{w}/p/q")
    os.makedirs(f"{w}/include")
    shutil.copy(f"{ROOT}/include/s5fxp.h", f"{w}/include/")
    for f in os.listdir(f"{ROOT}/sparsernns_amd/csrc"):
        shutil.copy(f"{ROOT}/sparsernns_amd/csrc/{f}", f"{w}/p/q/")
    for i in range(0, len(rest), 3):
        f, old, new = rest[i:i + 3]
        p = f"{w}/p/q/{f}"
        s = open(p).read()
        assert s.count(old) == 1, f"{name}: {f}: {s.count(old)} matches of {old!r}"
        open(p, "w").write(s.replace(old, new))
    os.makedirs(f"{ROOT}/tools/bin/{name}", exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-Wno-unused-value"] + flags + ["-o", f"{ROOT}/tools/bin/{name}/libs5fxp.so", "s5fxp_api.hip"],
                          cwd=f"{w}/p/q")
    print("built", name)


if __name__ == "__main__":
    main()
