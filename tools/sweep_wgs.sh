#!/bin/bash
# sweep_wgs.sh: per-kernel launch durations (rocprofv3, one forward at a time) for different workgroup counts of one kernel
#   tools/sweep_wgs.sh ENC 256 384 512 768   (ENC | DEC | CGATE | BPROJ | RESID)
k=$1; shift
for n in "$@"; do
  export S5FXP_WGS_$k=$n
  tools/prof_stats.sh sweep_${k}_$n --steps 10 --warmup 2 --no-cpu-baseline --no-scan-sweep --inflight 1 > /dev/null
  echo "S5FXP_WGS_$k=$n: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep_${k}_$n.json | head -1)"
  grep "k_enc\|k_dec\|k_cgate\|k_bproj\|k_resid" gpurun_out/sweep_${k}_$n.txt | cut -c1-40,60-100
  unset S5FXP_WGS_$k
done
