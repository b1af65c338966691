#!/bin/bash
# run_inflight.sh NAME...: bench (3 in flight + single pass) with each tools/bin/NAME/libs5fxp.so ("base" = in-tree)
for v in "$@"; do
  if [ "$v" = base ]; then unset S5FXP_LIB; else export S5FXP_LIB=$PWD/tools/bin/$v/libs5fxp.so; fi
  echo "== $v $(python3 bench.py --steps 36 --warmup 6 --no-cpu-baseline --no-scan-sweep | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
