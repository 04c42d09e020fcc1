#!/bin/bash
# gpurun_out/prof_<round>_<workload>/{summary.txt, trace/*kernel_stats.csv} -> profiles/<round>_<workload>_{rocprofv3_summary.txt, kernel_stats.csv}
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
ROUND=${1:-r06}
for d in $R/gpurun_out/prof_${ROUND}_*; do
  [ -d "$d" ] || continue
  w=$(basename $d | sed "s/^prof_${ROUND}_//")
  [ -f $d/summary.txt ] && cp $d/summary.txt $R/profiles/${ROUND}_${w}_rocprofv3_summary.txt
  f=$(find $d/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/profiles/${ROUND}_${w}_kernel_stats.csv
done
ls $R/profiles | grep ${ROUND}_
