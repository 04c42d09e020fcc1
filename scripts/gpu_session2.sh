#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s2}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in and_w8.so and_abl1.so and_abl2.so and_abl3.so; do
  echo "== and3 lib=$w" | tee -a $OUT/session.log
  RUCENE_GPU_LIB=$R/build_variants/$w timeout 300 python scripts/run_workload.py and3 5 2>&1 | tail -1 | tee -a $OUT/session.log
done
echo "== counters" | tee -a $OUT/session.log
RUCENE_GPU_LIB=$R/build_variants/expcount.so timeout 300 python scripts/run_workload.py and3 3 2>&1 | tail -2 | tee -a $OUT/session.log
echo "== pmc and3 (w8)" | tee -a $OUT/session.log
export RUCENE_GPU_LIB=$R/build_variants/and_w8.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $R/scripts/run_workload.py and3 2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- python $R/scripts/run_workload.py and3 2 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --output-format csv -d $OUT/pmc3 -o p -- python $R/scripts/run_workload.py and3 2 > $OUT/pmc3.log 2>&1
cd $R
python scripts/summarize_prof.py $OUT 2>&1 | grep -E "==|k_search_and" | tee -a $OUT/session.log
