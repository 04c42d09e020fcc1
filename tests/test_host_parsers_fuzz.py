"""Memory-safety hardening of the host-side file readers (rucene_amd/csrc/host/*.hpp: header-only C++, no HIP): a
mutation fuzzer built with AddressSanitizer + UndefinedBehaviorSanitizer damages valid files (written by the oracle and the
synthetic index writer) in thousands of ways and runs every parser. A reader may accept or reject damaged bytes, but it must
never read out of bounds, overflow or crash — those files come from disk."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_files(oracle, tmp_path):
    import __graft_entry__ as g
    g.build()
    from rucene_amd import indexgen
    from test_index_directory import build_directory
    build_directory(oracle, str(tmp_path), sizes=((20_000, 1_500),))
    exe = str(tmp_path / "fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_parsers_fuzz.cpp")])
    d = str(tmp_path)
    files = [os.path.join(d, f) for f in ("_0_Lucene50_0.doc", "_0_Lucene50_0.tim", "_0_Lucene50_0.tip", "_0.nvm", "_0.nvd", "_0_2.liv",
                                           "_0.fnm", "_0.si", "segments_2")]
    assert all(os.path.exists(f) for f in files)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2):
        r = subprocess.run([exe, "2500", str(seed)] + files, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        rejected = {line.split()[0]: int(line.split()[-1]) for line in r.stdout.strip().splitlines()}
        assert all(v > 0 for v in rejected.values()), rejected     # every parser saw damage it refused
