#!/bin/bash
# ab_bench.sh REPS VARIANT...: bench.py (no CPU baseline) for the shipped library ("base") and for
# tools/bin/VARIANT/libs5fxp.so, interleaved REPS times: value, ms per step in flight and one at a time
root=${GRAFT_REPO_ROOT:-$PWD}
reps=$1; shift
for rep in $(seq $reps); do
  for v in base "$@"; do
    if [ $v = base ]; then lib=""; else lib=$root/tools/bin/$v/libs5fxp.so; fi
    S5FXP_LIB=$lib python $root/bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-scan-sweep 2>/dev/null | tail -1 > /tmp/ab_$v.json
    python - $v <<'P'
import json, sys
d = json.load(open(f"/tmp/ab_{sys.argv[1]}.json"))
print(f"{sys.argv[1]:14s} value {d['value']:.4g}  in flight {d['ms_per_step']:.4f} ms  single {d['single_stream']['ms_per_step']:.4f} ms  scan {d['roofline']['avg_kernel_us']:.2f} us")
P
  done
done
