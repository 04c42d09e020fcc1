#!/bin/bash
# The FETCH_SIZE calibration of profiles/r06_fetch_granule.txt (scripts/microbench/fetch_granule.hip under rocprofv3 --pmc FETCH_SIZE), behind the GPU suite
# and the batch-of-one probe. usage (GPU box): bash scripts/fetch_granule_run.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-fetch_granule}; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python scripts/latency_probe.py --kinds or10,and3,term --calls 64 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/latency.txt
cd /tmp && export TMPDIR=/tmp
$R/scripts/microbench/fetch_granule > $OUT/fetch_granule.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fg -o p -- $R/scripts/microbench/fetch_granule > $OUT/fg.log 2>&1
cat $OUT/fetch_granule.txt
find $OUT/fg -name "*counter_collection.csv" | head -2
python - "$OUT" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
for path in glob.glob(out + "/fg/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == "FETCH_SIZE":
            acc[row["Kernel_Name"][:40]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("FETCH_SIZE %-40s dispatches %d  last %.1f KiB = %.1f MiB" % (k, len(v), v[-1], v[-1] / 1024.0))
P
