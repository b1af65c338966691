#!/bin/bash
# ab_gate.sh: N2 A/B.  The gate kernel with the test an activation-gating kernel makes before every MFMA of
# its two projections (tools/variant.py gate_count -DS5_GATE_CHECK=1 for the counters, gate_skip -DS5_GATE_CHECK=2 for the timing), against the shipped kernel, on configs[3]'s model at 32 x 4096
# per launch (--config 4 is the dense w4a8 dim 1.0 model; configs[3]'s own bench line is one 512-sequence batch) and on
# configs[1]; then the counters: how many fragments were all zero.
set -e
S5FXP_LIB=$PWD/tools/bin/gate_count/libs5fxp.so python3 tools/gate_counts.py
BENCH_ARGS="--config 3 --batch 64" bash tools/run_variants.sh base gate_skip 2>&1 | grep -E "^==|cgate"
BENCH_ARGS="" bash tools/run_variants.sh base gate_skip 2>&1 | grep -E "^==|cgate"
