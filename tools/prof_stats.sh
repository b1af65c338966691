#!/bin/bash
# prof_stats.sh NAME [bench.py args...]: rocprofv3 --kernel-trace --stats of bench.py; compact summary in gpurun_out/NAME.txt,
# the stats csv in gpurun_out/NAME_kernel_stats.csv (copy the ones to be judged into profiles/)
export TMPDIR=/tmp
name=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$name
rm -rf $out; mkdir -p $out
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py "$@" > $out.json 2> $out.err ) || { echo "$name FAILED"; tail -5 $out.err; exit 1; }
python3 $root/tools/kstats.py $out > $out.txt
cp $(ls $out/*/*kernel_stats.csv | head -1) ${out}_kernel_stats.csv
rm -rf $out
echo "== $name $(grep -o '"ms_per_step": [0-9.]*' $out.json | head -1)"; head -12 $out.txt
