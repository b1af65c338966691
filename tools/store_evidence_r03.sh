#!/bin/bash
# store_evidence_r03.sh: gpurun_out/ (tools/evidence_r03.sh) -> profiles/r03_v3_*, profiles/r03_scan_traffic.json
cd gpurun_out || exit 1
for f in r03_g1_single r03_g8_single r03_default; do cp $f.txt ../profiles/r03_v3_${f#r03_}_kernel_stats.txt; cp ${f}_rocprofv3_kernel_stats.csv ../profiles/r03_v3_${f#r03_}_rocprofv3_kernel_stats.csv; done
for f in r03_pmc_fetch_g1 r03_pmc_write_g1 r03_pmc_fetch_g8 r03_pmc_write_g8 r03_pmc_sq_g8; do cp $f.txt ../profiles/r03_v3_${f#r03_}.txt; done
cd ..
python3 tools/make_traffic_json.py gpurun_out/r03 profiles/r03_scan_traffic.json
cp gpurun_out/bench_v3.json profiles/r03_v3_bench.json
for c in 2 3 4; do cp gpurun_out/bench_v3_c$c.json profiles/r03_v3_bench_config$c.json; done
