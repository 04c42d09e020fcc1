set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g1/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g1/pytest.log
for w in term decode and3 or10; do timeout 300 python scripts/run_workload.py $w 5 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g1/pmc -o p -- python $GRAFT_REPO_ROOT/scripts/run_workload.py term 2 > $GRAFT_REPO_ROOT/gpurun_out/g1/pmc.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/g1/pmc/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'search_term' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
