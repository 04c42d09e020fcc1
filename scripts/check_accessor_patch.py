#!/usr/bin/env python3
"""rust/rucene_accessors.patch against the crate it patches — run in the container that holds the reference checkout
(usage: check_accessor_patch.py [/path/to/rucene]; default /root/reference). Checks that every hunk's anchor line number holds the
`impl` line its header names and that `patch -p1 --dry-run` accepts the file on a scratch copy of the three sources."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
path = os.path.join(ROOT, "rust", "rucene_accessors.patch")
files, cur = {}, None
for line in open(path).read().split("\n"):
    if line.startswith("+++ b/"):
        cur = line[6:].strip()
    m = re.match(r"@@ -(\d+),0 \+(\d+),(\d+) @@ (.*)", line)
    if m:
        files[cur] = (int(m.group(1)), m.group(4).strip())
with tempfile.TemporaryDirectory() as tmp:
    for rel, (n, anchor) in files.items():
        src = open(os.path.join(ref, rel)).read().split("\n")
        assert src[n - 1].strip() == anchor, (rel, n, src[n - 1])
        os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
        shutil.copy(os.path.join(ref, rel), os.path.join(tmp, rel))
        print("anchor ok:", rel, n, anchor)
    subprocess.run(["patch", "-p1", "--dry-run", "-s", "-i", path], cwd=tmp, check=True)
print("patch --dry-run: ok")
