#!/bin/bash
# round 5, GPU call 5: chunks of workgroups dealt to the XCDs (one L2 per list instead of eight), with and without the denser probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c5; mkdir -p $OUT
cd $R
run() {  # lib workload docs [env...]
  local lib=$1 w=$2 docs=$3; shift 3
  if [ "$lib" = "default" ]; then unset RUCENE_GPU_LIB; else export RUCENE_GPU_LIB=$R/build_variants/$lib.so; fi
  echo "== $w docs=$docs lib=$lib $*" | tee -a $OUT/ab.log
  env "$@" DOCS=$docs timeout 600 python scripts/run_workload.py $w 8 2>&1 | grep "k_search_\|k_prepare_blocks" | tail -1 | grep -o "'k_search_and[^)]*)\|'k_search_term[^)]*)\|'k_merge_items[^)]*)\|'k_phrase_match_lanes[^)]*)" | tr '\n' ' ' | tee -a $OUT/ab.log; echo | tee -a $OUT/ab.log
}
for rep in 1 2; do
  for lib in default and_xcd4 and_xcd16 and_xcd64 and_xcd256 and_xcd16w and_xcd64w; do run $lib and3 10000000 X=1; done
done
for lib in default and_xcd16 and_xcd64 and_xcd256 and_xcd64w; do run $lib and3 100000000 X=1; done
run and_xcd64 phrase2 10000000 X=1
export RUCENE_GPU_LIB=$R/build_variants/and_xcd64.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conjunction or lazy or must or phrase or filter or nested" > $OUT/pytest.log 2>&1; echo "pytest(xcd64) rc=$?" | tee -a $OUT/ab.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3 | tee -a $OUT/ab.log
unset RUCENE_GPU_LIB
cd /tmp && export TMPDIR=/tmp
RUCENE_GPU_LIB=$R/build_variants/and_xcd64.so timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_xcd64 -o p -- python $R/scripts/run_workload.py and3 2 > $OUT/pmc_xcd64.log 2>&1
cd $R
python - <<'PY' 2>&1 | tee -a gpurun_out/r5c5/ab.log
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5c5/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "k_search_and" in row.get("Kernel_Name", ""):
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    print(f.split("/")[2], {k: v[0] / max(1, v[1]) for k, v in acc.items()})
PY
