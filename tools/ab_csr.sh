#!/bin/bash
# ab_csr.sh: N1 A/B.  B projection with the weight operand as compressed columns and a VALU
# contraction (tools/variant.py bproj_csr16 -DS5_BPROJ_CSR=16) against the shipped dense zero-filled MFMA operand, on
# BASELINE configs[2] (90 % pruned weights).  First the variant's parity (bench.py's CPU leg compares the outputs), then
# the kernel durations of both builds.
set -e
echo "== parity of the CSR build (config 2)"
S5FXP_LIB=$PWD/tools/bin/bproj_csr16/libs5fxp.so python3 bench.py --config 2 --steps 16 --warmup 2 --no-scan-sweep --cpu-seconds 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('matches_gpu', d['cpu_baseline'].get('matches_gpu'), 'value %.4g' % d['value'])"
BENCH_ARGS="--config 2" bash tools/run_variants.sh base bproj_csr16
