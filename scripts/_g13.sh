cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 3 --force-dist --no-cpu-baseline > /tmp/o.txt 2> /tmp/e.txt; echo rc=$?
echo "stdout lines: $(wc -l < /tmp/o.txt)"; cut -c1-150 /tmp/o.txt; grep -c "ROCm version\|force-dist" /tmp/e.txt
timeout 600 python bench.py > /tmp/o2.txt 2> /tmp/e2.txt; echo rc=$?; echo "stdout lines: $(wc -l < /tmp/o2.txt)"; python -c "
import json; d=json.loads(open('/tmp/o2.txt').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_vs_oracle_full_batch'])"
