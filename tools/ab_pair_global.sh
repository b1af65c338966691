for env in "A=1" "S5FXP_PAIR_GLOBAL=1"; do for infl in 1 3; do echo -n "$env inflight $infl groups 1: "; env $env python bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-scan-sweep --groups 1 --inflight $infl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(f\"value {d['value']:.4g} ms {d['ms_per_step']} scan {r['avg_kernel_us']} frac {r['frac']} kernel {r['kernel'][:18]}\")"; done; done
echo -n "PAIR_GLOBAL groups 8: "; S5FXP_PAIR_GLOBAL=1 python bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-scan-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(f\"value {d['value']:.4g} ms {d['ms_per_step']} scan {r['avg_kernel_us']} one {r['one_batch_launch']['avg_kernel_us']} {r['one_batch_launch']['frac']}\")"
