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
    # a second dictionary whose field is indexed with positions + offsets + payloads (three file pointers per term)
    rng = np.random.default_rng(4)
    n = 1200
    st = np.zeros(n, dtype=oracle.FULL_TERM_STATE_DTYPE)
    df = rng.choice([1, 2, 100, 129, 4000], size=n)
    st["base"]["doc_freq"] = df
    st["base"]["total_term_freq"] = df + rng.integers(0, 300, n)
    st["base"]["doc_start_fp"] = 40 + np.cumsum(rng.integers(0, 900, n))
    st["base"]["singleton_doc_id"] = np.where(df == 1, 77, -1)
    st["base"]["skip_offset"] = np.where(df > 128, 999, -1)
    st["pos_start_fp"] = np.cumsum(rng.integers(0, 5000, n))
    st["pay_start_fp"] = np.cumsum(rng.integers(0, 300, n))
    st["last_pos_block_offset"] = np.where(st["base"]["total_term_freq"] > 128, 12345, -1)
    ptim, ptip = oracle.blocktree_write([dict(number=1, index_options=4, has_payloads=True, doc_count=1, terms=[b"w%05d" % i for i in range(n)],
                                              states=st)], 5, 10)
    for name, data in (("pos.tim", ptim), ("pos.tip", ptip)):
        with open(os.path.join(d, name), "wb") as fh:
            fh.write(data)
    files += [os.path.join(d, "pos.tim"), os.path.join(d, "pos.tip")]
    # a compound file holding the small files of the segment
    sid = open(os.path.join(d, "_0.fnm"), "rb").read()[4 + 1 + len("Lucene60FieldInfos") + 4:][:16]
    cfs, cfe = oracle.compound_write({n: open(os.path.join(d, n), "rb").read() for n in ("_0.fnm", "_0.nvm", "_0.nvd", "_0_Lucene50_0.tip")}, sid)
    for name, data in (("_0.cfe", cfe), ("_0.cfs", cfs)):
        with open(os.path.join(d, name), "wb") as fh:
            fh.write(data)
    files += [os.path.join(d, "_0.cfe"), os.path.join(d, "_0.cfs")]
    assert all(os.path.exists(f) for f in files)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2):
        r = subprocess.run([exe, "2500", str(seed)] + files, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        rejected = {line.split()[0]: int(line.split()[-1]) for line in r.stdout.strip().splitlines()}
        assert all(v > 0 for v in rejected.values()), rejected     # every parser saw damage it refused
