#!/bin/bash
# kres_build.sh [regex]: compile the library with --save-temps in /tmp and print per-kernel resources
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/st && mkdir -p /tmp/st/p/q /tmp/st/include && cp $root/include/s5fxp.h /tmp/st/include && cp $root/sparsernns_amd/csrc/* /tmp/st/p/q/
cd /tmp/st/p/q && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value --save-temps=obj -o /tmp/st/lib.so s5fxp_api.hip 2>&1 | grep -E "error" | head
python3 $root/tools/kres.py /tmp/st/s5fxp_api-hip-amdgcn-amd-amdhsa-gfx950.s "$1"
