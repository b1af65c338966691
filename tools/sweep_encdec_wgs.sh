#!/bin/bash
# sweep_encdec_wgs.sh: encoder / decoder workgroups per launch (S5FXP_WGS_ENC / _DEC), kernel durations by rocprofv3, 8 batches per launch
for w in 512 256 1024 2048; do export S5FXP_WGS_DEC=$w S5FXP_WGS_ENC=$w; echo "== WGS_DEC=WGS_ENC=$w"; BENCH_ARGS="--steps 48 --no-one-batch-pass" bash tools/run_variants.sh base 2>&1 | grep -E "enc_p|dec_p"; done
