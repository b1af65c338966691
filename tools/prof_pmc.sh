#!/bin/bash
# prof_pmc.sh NAME "CTR1 CTR2 ..." [bench.py args...]: one rocprofv3 --pmc pass over bench.py (counters only: no trace flags
# beside them); per-kernel means in gpurun_out/NAME.txt
export TMPDIR=/tmp
name=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$name
rm -rf $out; mkdir -p $out
( cd /tmp && rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $root/bench.py "$@" > $out.json 2> $out.err ) || { echo "$name FAILED"; tail -5 $out.err; exit 1; }
python3 $root/tools/pmc_summary.py $out > $out.txt
rm -rf $out
echo "== $name ($ctrs)"; cat $out.txt
