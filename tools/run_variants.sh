#!/bin/bash
# run_variants.sh NAME...: kernel-trace the bench with each tools/bin/NAME/libs5fxp.so ("base" = the in-tree build);
# BENCH_ARGS adds bench.py arguments
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset S5FXP_LIB; else export S5FXP_LIB=$PWD/tools/bin/$v/libs5fxp.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_$v -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-scan-sweep --inflight 1 $BENCH_ARGS > gpurun_out/var_$v.log 2>&1 || { echo "$v FAILED"; tail -5 gpurun_out/var_$v.log; }
  echo "== $v $(grep -o 'ms_per_step[^,]*' gpurun_out/var_$v.log | head -1)"
  python3 tools/kstats.py gpurun_out/var_$v > gpurun_out/var_$v.txt; grep -E "enc_p|bproj_p|cgate_p|resid_minmax|dec_p|scan_pair" gpurun_out/var_$v.txt
  rm -rf gpurun_out/var_$v
done
