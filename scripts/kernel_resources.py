#!/usr/bin/env python3
"""Compile rgpu_api.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs / SGPRs / scratch bytes per lane / occupancy / LDS. usage: kernel_resources.py [-DFLAG ...] [-o lib.so]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
out = "/tmp/kernel_resources.so"
if "-o" in args:
    i = args.index("-o")
    out = args[i + 1]
    del args[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
       "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-o", out,
       os.path.join(ROOT, "rucene_amd", "csrc", "rgpu_api.hip"), "-L/opt/rocm/lib", "-lrccl"] + args
p = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
if p.returncode != 0:
    sys.stderr.write(p.stderr)
    sys.exit(p.returncode)
cur = None
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void rgpu::", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("spill_v", r"VGPRs Spill: (\d+)"), ("spill_s", r"SGPRs Spill: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-60s %5s %5s %8s %4s %7s %7s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "occ", "lds", "spillV", "spillS"))
for r in rows:
    print("%-60s %5s %5s %8s %4s %7s %7s %7s" % (r["name"][:60], r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("occ"),
                                                r.get("lds"), r.get("spill_v"), r.get("spill_s")))
