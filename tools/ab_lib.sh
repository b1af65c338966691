#!/bin/bash
# ab_lib.sh NAME [bench args]: A/B of tools/bin/NAME/libs5fxp.so against the in-tree build on one box, alternating
# (base, NAME, base, NAME): whole-bench numbers, sustained (not rocprof) so that the clocks are the bench's
n=$1; shift
for rep in 1 2; do for v in base $n; do
  if [ "$v" = base ]; then unset S5FXP_LIB; else export S5FXP_LIB=$PWD/tools/bin/$v/libs5fxp.so; fi
  echo -n "$v: "; python3 bench.py --no-cpu-baseline --no-scan-sweep "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('single_set') or {}
print('value %.4g ms %.4f scan %.1f' % (d['value'], d['ms_per_step'], r['avg_kernel_us']), {k:v for k,v in d.items() if k in ('one_set_at_a_time',)})"
done; done
