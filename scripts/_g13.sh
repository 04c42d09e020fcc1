cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/g13
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/g13/pmc -o p -- python $R/scripts/run_workload.py term 2 > $R/gpurun_out/g13/pmc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/g13/pmc2 -o p -- python $R/scripts/run_workload.py term 2 > $R/gpurun_out/g13/pmc2.log 2>&1
cd $R; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/g13/pmc*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'search_term' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
tail -2 gpurun_out/g13/pmc2.log | cut -c1-200
