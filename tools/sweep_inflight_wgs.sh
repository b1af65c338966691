#!/bin/bash
# sweep_inflight_wgs.sh: bench.py (three batches in flight, no CPU leg) with different workgroup caps of the tile kernels
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 120 --warmup 12 --no-cpu-baseline --no-scan-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"value {d['value']:.4g} inflight {d['ms_per_step']:.4f} single {d['single_stream']['ms_per_step']:.4f}\")"
}
run A=1
run S5FXP_WGS_ENC=256 S5FXP_WGS_DEC=256 S5FXP_WGS_CGATE=256 S5FXP_WGS_BPROJ=512 S5FXP_WGS_RESID=256
run S5FXP_WGS_ENC=256
run S5FXP_WGS_CGATE=256
run S5FXP_WGS_DEC=256
run S5FXP_WGS_BPROJ=512
run S5FXP_WGS_RESID=256
run S5FXP_WGS_ENC=384 S5FXP_WGS_DEC=384 S5FXP_WGS_CGATE=384 S5FXP_WGS_BPROJ=768 S5FXP_WGS_RESID=384
run S5FXP_WGS_ENC=768 S5FXP_WGS_DEC=768 S5FXP_WGS_CGATE=768 S5FXP_WGS_BPROJ=2048 S5FXP_WGS_RESID=1024
run S5FXP_NO_CGATE_DMA=1
run A=1
