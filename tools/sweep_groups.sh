#!/bin/bash
# sweep_groups.sh: bench.py with G batches per launch set and different per-group workgroup caps
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 120 --warmup 12 --no-cpu-baseline --no-scan-sweep $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"value {d['value']:.4g} inflight {d['ms_per_step']:.4f} single {d['single_stream']['ms_per_step']:.4f}\")"
}
ARGS="--groups 4"
run A=1
run S5FXP_WGS_ENC=128 S5FXP_WGS_DEC=128 S5FXP_WGS_CGATE=128 S5FXP_WGS_BPROJ=256 S5FXP_WGS_RESID=128
run S5FXP_WGS_ENC=256 S5FXP_WGS_DEC=256 S5FXP_WGS_CGATE=256 S5FXP_WGS_BPROJ=512 S5FXP_WGS_RESID=256
run S5FXP_WGS_ENC=192 S5FXP_WGS_DEC=192 S5FXP_WGS_CGATE=192 S5FXP_WGS_BPROJ=384 S5FXP_WGS_RESID=192
ARGS="--groups 8"
run S5FXP_WGS_ENC=64 S5FXP_WGS_DEC=64 S5FXP_WGS_CGATE=64 S5FXP_WGS_BPROJ=128 S5FXP_WGS_RESID=64
run S5FXP_WGS_ENC=128 S5FXP_WGS_DEC=128 S5FXP_WGS_CGATE=128 S5FXP_WGS_BPROJ=256 S5FXP_WGS_RESID=128
ARGS="--groups 4 --inflight 2"
run S5FXP_WGS_ENC=128 S5FXP_WGS_DEC=128 S5FXP_WGS_CGATE=128 S5FXP_WGS_BPROJ=256 S5FXP_WGS_RESID=128
ARGS="--groups 8 --inflight 2"
run S5FXP_WGS_ENC=64 S5FXP_WGS_DEC=64 S5FXP_WGS_CGATE=64 S5FXP_WGS_BPROJ=128 S5FXP_WGS_RESID=64
