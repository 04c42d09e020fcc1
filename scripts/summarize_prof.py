"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary per kernel."""
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for p in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(p):
        continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:48]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
        lines.append("== %s (mean per dispatch)" % os.path.relpath(f, out))
        for k in agg:
            lines.append("  %-48s %s" % (k, "  ".join("%s=%.4g" % (c, agg[k][c] / cnt[k][c]) for c in sorted(agg[k]))))
txt = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
print(txt)
