#!/bin/bash
# build_variant.sh NAME ['sed-expr' FILE ...]: builds tools/bin/NAME/libs5fxp.so from a (patched) copy of csrc; EXTRA_FLAGS are
# passed to hipcc (e.g. EXTRA_FLAGS=-DS5_PHASE_PROF)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/v/$name; rm -rf $w; mkdir -p $w/p/q $w/include $root/tools/bin/$name
cp $root/include/s5fxp.h $w/include/; cp $root/sparsernns_amd/csrc/* $w/p/q/
while [ $# -gt 1 ]; do sed -i "$1" $w/p/q/$2; shift 2; done
cd $w/p/q && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $EXTRA_FLAGS \
   -o $root/tools/bin/$name/libs5fxp.so s5fxp_api.hip
