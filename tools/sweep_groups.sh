#!/bin/bash
# sweep_groups.sh: bench.py with G batches per launch set and different numbers of sets in flight
run() {
  echo -n "$* : "
  python bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-scan-sweep "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"value {d['value']:.4g} inflight {d['ms_per_step']:.4f} single {d['single_stream']['ms_per_step']:.4f}\")"
}
run --groups 1
run --groups 4
run --groups 8
run --groups 16
run --groups 8 --inflight 2
run --groups 16 --inflight 2
run --groups 8 --inflight 4
run --groups 8 --seq-len 3751
