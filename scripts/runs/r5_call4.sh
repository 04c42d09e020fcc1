#!/bin/bash
# round 5, GPU call 4: what bounds the first probe's gathers (cache policy / structure / all-hit ablation), TERM item sizes, cold-path host timing, L1/L2 counters
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c4; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default and_nt and_words and_abl7 and_g8w3; do run $lib and3 10000000 X=1; done
done
for b in 0 64 128 256 1024; do run default term 10000000 BPI=$b; done
for lib in default and_nt and_words; do run $lib and3 100000000 X=1; done
unset RUCENE_GPU_LIB
echo "== cold 100M host timing" | tee -a $OUT/ab.log
RGPU_HOST_TIMING=1 DOCS=100000000 timeout 600 python scripts/run_workload.py cold 3 2>&1 | grep -i "prepare host\|footprint\|k_prepare" | cut -c1-600 | tee -a $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -o p -- python $R/scripts/run_workload.py and3 2 > $OUT/pmc_$tag.log 2>&1
  echo "pmc $set rc=$?" | tee -a $OUT/ab.log
done
cd $R
python - <<'PY' 2>&1 | tee -a gpurun_out/r5c4/ab.log
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5c4/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "k_search_and" in row.get("Kernel_Name", ""):
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    print(f.split("/")[2], {k: v[0] / max(1, v[1]) for k, v in acc.items()})
PY
