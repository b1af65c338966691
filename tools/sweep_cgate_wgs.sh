#!/bin/bash
# sweep_cgate_wgs.sh: gate-kernel workgroups per launch (S5FXP_WGS_CGATE), kernel durations by rocprofv3, 8 batches per launch
for w in 512 256 384 768 1024; do export S5FXP_WGS_CGATE=$w; echo "== WGS_CGATE=$w"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "cgate_p"; done
