set -u
cd $GRAFT_REPO_ROOT
echo base; for w in term decode and3 or10; do timeout 300 python scripts/run_workload.py $w 5 2>&1 | tail -1; done
for v in w4 w8; do for b in 0 128 256; do echo $v BPI=$b; BPI=$b RUCENE_GPU_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so timeout 300 python scripts/run_workload.py term 5 2>&1 | tail -1; done; done
echo base BPI=128;  BPI=128 timeout 300 python scripts/run_workload.py term 5 2>&1 | tail -1
echo base BPI=256;  BPI=256 timeout 300 python scripts/run_workload.py term 5 2>&1 | tail -1
