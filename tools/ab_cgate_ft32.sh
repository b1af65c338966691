#!/bin/bash
# ab_cgate_ft32.sh: the gate kernel on 32-frame tiles with three-wave workgroups (S5FXP_CGATE_FT32=1): with the sigmoid table sized
# exactly, five of them fit a CU's LDS and registers (15 waves instead of 12).  Parity first, then kernel durations per grid size
# (the 32-frame grid is 2 x S5FXP_WGS_CGATE), then the shipped 64-frame kernel.
export S5FXP_CGATE_FT32=1
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_forward_matches_oracle or any_sequence_length or grouped or live_states or decoder_carries" 2>&1 | tail -1
for w in 512 640 768 1024; do export S5FXP_WGS_CGATE=$w; echo "== FT32, $((2 * w)) workgroups per launch"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "cgate_p"; done
unset S5FXP_CGATE_FT32 S5FXP_WGS_CGATE
echo "== FT64 (shipped)"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "cgate_p"
