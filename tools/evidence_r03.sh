#!/bin/bash
# evidence_r03.sh: everything DESIGN.md section 6 quotes, on one GPU box: tools/prof_r03.sh (kernel stats in three modes + PMC passes) and the
# bench lines of configs[1]-[4] with the CPU leg (config 3 with its whole 512-sequence batch on the CPU side).  Results under gpurun_out/;
# copy to profiles/ with tools/store_evidence_r03.sh.
bash tools/prof_r03.sh > gpurun_out/prof.log 2>&1
grep -E "^== r03_(g1|g8|default)" -A7 gpurun_out/prof.log | grep -E "^==|scan_pairl|cgate_p|bproj_p|dec_p|enc_p|resid"
python3 bench.py > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err
for c in 2 4; do python3 bench.py --config $c > gpurun_out/bench_v3_c$c.json 2> gpurun_out/bench_v3_c$c.err || echo "config $c failed"; done
python3 bench.py --config 3 --cpu-batch 512 --cpu-seconds 20 > gpurun_out/bench_v3_c3.json 2> gpurun_out/bench_v3_c3.err || echo "config 3 failed"
python3 - <<PY
import json
for f in ("bench_v3","bench_v3_c2","bench_v3_c3","bench_v3_c4"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}; g=d.get("gate_kernel") or {}
    print(f, "%.4g" % d["value"], d["ms_per_step"], "single", (d.get("single_stream") or {}).get("ms_per_step"), "scan", r.get("avg_kernel_us"), r.get("frac"), "gate", g.get("avg_kernel_us"), g.get("frac_moved"), "cpu %.3g" % c.get("value",0), c.get("matches_gpu"))
PY
