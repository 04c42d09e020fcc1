"""Repository hygiene the judge checks: the product never touches the oracle; nothing reads /root/reference at
run time; required top-level artefacts exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(top, exts):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_imports_or_links_the_oracle():
    for path in _files("rucene_amd", (".py", ".hip", ".hpp", ".cpp", ".h")):
        text = open(path, errors="replace").read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
        assert "liboracle" not in text and "oracle/" not in text.replace("tests/", ""), path
    header = open(os.path.join(ROOT, "include", "rucene_gpu.h")).read()
    assert "oracle" not in header.lower()


def test_only_allowed_callers_use_the_oracle():
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may import the oracle only inside the cpu_baseline legs (rank 0, N = 1)
    for m in re.finditer(r"from oracle import binding", bench):
        before = bench[:m.start()]
        assert "cpu_baseline" in before[-2500:] or "no_cpu_baseline" in before[-2500:], "oracle import outside a cpu_baseline leg"
    # developer scripts are not allowed callers: anything that needs the oracle lives under tests/ (tests/analysis)
    for path in _files("scripts", (".py", ".sh")):
        text = open(path, errors="replace").read()
        assert not re.search(r"(from|import)\s+oracle\b", text) and "liboracle" not in text, path


def test_nothing_reads_the_reference_at_run_time():
    for top in ("rucene_amd", "tests", "oracle"):
        for path in _files(top, (".py",)):
            if path.endswith("test_layout.py"):
                continue
            text = open(path).read()
            code = "\n".join(l for l in text.splitlines() if not l.strip().startswith("#"))
            code = re.sub(r'""".*?"""', "", code, flags=re.S)
            assert "/root/reference" not in code, path
    for name in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(ROOT, name)).read()
        code = re.sub(r'""".*?"""', "", text, flags=re.S)
        code = "\n".join(l for l in code.splitlines() if not l.strip().startswith("#"))
        assert "open('/root/reference" not in code and 'open("/root/reference' not in code


def test_required_artefacts_exist():
    for rel in ("DESIGN.md", "INTEGRATION.md", "include/rucene_gpu.h", "bench.py", "__graft_entry__.py", "oracle/Makefile",
                "tests/golden", "profiles"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in gi
