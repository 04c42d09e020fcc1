set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
for v in noslow; do echo $v; for b in 0 64 256; do BPI=$b RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so timeout 300 python scripts/run_workload.py term 5 2>&1 | tail -1; done; done
cd /tmp && export TMPDIR=/tmp
for v in noslow; do
RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/g3/pmc_$v -o p -- python $GRAFT_REPO_ROOT/scripts/run_workload.py term 2 > $GRAFT_REPO_ROOT/gpurun_out/g3/pmc_$v.log 2>&1
done
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/g3/pmc_*/**/*counter_collection.csv', recursive=True):
    print(f)
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'search_term' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
