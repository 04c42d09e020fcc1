#!/bin/bash
# Where one headline step's time goes ON THE GPU's clock: the H2D copy of the staged plan and the kernel(s), back to back on one stream.
# usage (GPU box): bash scripts/step_timeline.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-steptl}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python $R/scripts/host_call_probe.py > $OUT/probe.log 2>&1
cd $R
python - $OUT <<'P'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
tail = ev[-24:]
t0 = tail[0][0]
prev_end = t0
for s, e, n in tail:
    print("%9.1f us  +%6.1f  dur %6.1f  gap since previous end %6.1f  %s" % ((s - t0) / 1e3, 0, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = e
P
