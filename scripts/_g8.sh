cd $GRAFT_REPO_ROOT
for L in 1 2; do for S in 20 100; do
BENCH_LANES=$L timeout 600 python bench.py --steps $S --warmup 5 --force-dist --no-cpu-baseline > /tmp/o2.txt 2> /tmp/e2.txt; echo "dist lanes=$L steps=$S rc=$? $(grep -o '"ms_per_step": [0-9.]*' /tmp/o2.txt)"
done; done
for L in 1 2 3; do BENCH_LANES=$L timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > /tmp/o2.txt 2> /tmp/e2.txt; echo "local lanes=$L rc=$? $(grep -o '"ms_per_step": [0-9.]*' /tmp/o2.txt) $(grep -o '"k_search_term": [0-9.]*' /tmp/o2.txt)"; done
