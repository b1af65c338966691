#!/bin/bash
# sweep_groups2.sh: is G = 16 better because of its batches per launch or because of its workgroups per launch (the per-group
# floor of 64 workgroups makes 1024 per launch at G = 16)?
run() {
  echo -n "$* : "
  env $E python3 bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-scan-sweep "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"value {d['value']:.4g} inflight {d['ms_per_step']:.4f} single {d['single_stream']['ms_per_step']:.4f}\")"
}
E="A=1" run --groups 8
E="A=1" run --groups 16
E="S5FXP_WGS_ENC=1024 S5FXP_WGS_DEC=1024 S5FXP_WGS_CGATE=1024 S5FXP_WGS_BPROJ=2048 S5FXP_WGS_RESID=1024" run --groups 8
E="A=1" run --groups 24
E="A=1" run --groups 32
E="A=1" run --groups 16 --inflight 4
E="A=1" run --groups 8
E="A=1" run --groups 16
