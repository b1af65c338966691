#!/bin/bash
# prof_r03.sh: the round-3 evidence set on one GPU box.  rocprofv3 kernel stats of bench.py in three modes and the FETCH_SIZE /
# WRITE_SIZE passes (separate --pmc runs, counters only) for G = 1 and G = 8 batches per launch set; summaries in gpurun_out/.
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
stats() { # NAME bench-args...
  name=$1; shift
  rm -rf $out/$name; mkdir -p $out/$name
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 $root/bench.py "$@" > $out/$name.json 2> $out/$name.err ) || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python3 $root/tools/kstats.py $out/$name > $out/$name.txt
  cp $(ls $out/$name/*/*kernel_stats.csv | head -1) $out/${name}_rocprofv3_kernel_stats.csv
  rm -rf $out/$name
  echo "== $name $(grep -o '"ms_per_step": [0-9.]*' $out/$name.json | head -1)"; head -8 $out/$name.txt
}
pmc() { # NAME "CTRS" bench-args...
  name=$1; ctrs=$2; shift 2
  rm -rf $out/$name; mkdir -p $out/$name
  ( cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $out/$name -- python3 $root/bench.py "$@" > $out/$name.json 2> $out/$name.err ) || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python3 $root/tools/pmc_summary.py $out/$name > $out/$name.txt
  rm -rf $out/$name
  echo "== $name"; cat $out/$name.txt
}
common="--no-cpu-baseline --no-scan-sweep"
stats r03_g1_single --steps 24 --warmup 6 $common --inflight 1 --groups 1 &&
stats r03_g8_single --steps 48 --warmup 8 $common --inflight 1 --groups 8 --no-one-batch-pass &&
stats r03_default --steps 96 --warmup 24 $common --no-one-batch-pass &&
pmc r03_pmc_fetch_g1 "FETCH_SIZE" --steps 6 --warmup 2 $common --inflight 1 --groups 1 &&
pmc r03_pmc_write_g1 "WRITE_SIZE" --steps 6 --warmup 2 $common --inflight 1 --groups 1 &&
pmc r03_pmc_fetch_g8 "FETCH_SIZE" --steps 16 --warmup 8 $common --inflight 1 --groups 8 --no-one-batch-pass &&
pmc r03_pmc_write_g8 "WRITE_SIZE" --steps 16 --warmup 8 $common --inflight 1 --groups 8 --no-one-batch-pass &&
pmc r03_pmc_sq_g8 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS" --steps 16 --warmup 8 $common --inflight 1 --groups 8 --no-one-batch-pass
