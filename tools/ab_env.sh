#!/bin/bash
# ab_env.sh "ENV=1" [bench args]: whole-bench A/B of one environment switch on one box, alternating (off, on, off, on)
e=$1; shift
for rep in 1 2; do for v in off on; do
  echo -n "$e $v: "; if [ $v = on ]; then env $e python3 bench.py --no-cpu-baseline --no-scan-sweep "$@" 2>/dev/null; else python3 bench.py --no-cpu-baseline --no-scan-sweep "$@" 2>/dev/null; fi | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms %.4f' % (d['value'], d['ms_per_step']))"
done; done
