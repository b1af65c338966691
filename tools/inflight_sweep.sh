#!/bin/bash
# inflight_sweep.sh: frames/s of bench.py for 1..8 batches in flight (one line each)
for d in 1 2 3 4 5 6 8; do
  python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-scan-sweep --inflight $d 2>/dev/null | tail -1 > /tmp/sweep_$d.json
  python - $d <<'P'
import json, sys
d = json.load(open(f"/tmp/sweep_{sys.argv[1]}.json"))
print(f"batches in flight {sys.argv[1]}: {d['value']:.4g} frames/s  {d['ms_per_step']:.4f} ms per 32x4096 batch")
P
done
