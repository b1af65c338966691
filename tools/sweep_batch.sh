#!/bin/bash
# sweep_batch.sh: frames/s in flight and one at a time for several batch sizes (does the per-frame cost depend on the footprint?)
for b in 4 8 16 32 64 128; do
  python bench.py --steps 120 --warmup 12 --no-cpu-baseline --no-scan-sweep --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(f\"B=$b value {d['value']:.4g} inflight {d['ms_per_step']:.4f} ms ({d['ms_per_step']*1e6/($b*4096):.2f} ns/frame) single {d['single_stream']['ms_per_step']:.4f} ms ({d['single_stream']['ms_per_step']*1e6/($b*4096):.2f} ns/frame)\")"
done
