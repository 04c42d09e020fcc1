cd $GRAFT_REPO_ROOT
for v in and_base and_ad and_adc and_adb and_all and_ad_nodense and_all_nodense; do echo "== $v"; RUCENE_GPU_LIB=$PWD/build_variants/$v.so python scripts/run_workload.py and3 5 2>&1 | tail -1 | sed 's/.*k_search_and/k_search_and/' | cut -c1-120; done
