#!/bin/bash
# sweep_plane_skew.sh: does the distance between the workspace's planes matter (HBM channel mapping)?  S5FXP_PLANE_SKEW adds
# bytes to every plane; whole-bench numbers, two alternating repeats
for rep in 1 2; do for k in 0 4352 69888 1052928 2101504; do
  echo -n "skew $k: "; S5FXP_PLANE_SKEW=$k python3 bench.py --no-cpu-baseline --no-scan-sweep 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms %.4f' % (d['value'], d['ms_per_step']))"
done; done
