#!/bin/bash
# usage: scripts/prof.sh <workload> <tag>   (run on the GPU box through gpurun; writes gpurun_out/<tag>/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-term}; TAG=${2:-prof}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp WALL=0  # (WALL=0: run_workload.py without its untraced wall-clock leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/scripts/run_workload.py $W 5 > $OUT/trace.log 2>&1
if [ "${PROF_SHORT:-0}" != "1" ]; then  # (the 100 M-doc workloads take the trace and the two HBM passes only: every pass rebuilds the shard)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python $R/scripts/run_workload.py $W 2 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- python $R/scripts/run_workload.py $W 2 > $OUT/pmc2.log 2>&1
# (round 5) the vector L1 / L2 side of the gathers: L2 requests and hits, L1 -> L2 read requests, cycles the L1s sat on pending misses
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc5 -o p -- python $R/scripts/run_workload.py $W 2 > $OUT/pmc5.log 2>&1
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- python $R/scripts/run_workload.py $W 2 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- python $R/scripts/run_workload.py $W 2 > $OUT/pmc4.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
python $R/scripts/summarize_prof.py $OUT
