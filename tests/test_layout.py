"""CPU: repository contract -- the product package never touches oracle/ or the reference, has no CPU
fallback wording hiding one, and the required top-level pieces exist."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "temporalstereo_amd")


def _sources(top, exts):
    for d, _, files in os.walk(top):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_imports_oracle_or_reference():
    bad = []
    for path in _sources(PKG, (".py", ".hip", ".hpp")):
        txt = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "/root/reference" in txt:
            bad.append(path)
    assert not bad, bad


def test_runtime_files_do_not_read_reference():
    for rel in ("bench.py", "__graft_entry__.py", "benchlegs/common.py", "benchlegs/k1.py", "benchlegs/extras.py", "benchlegs/training.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, rel)).read()
    for path in _sources(os.path.join(ROOT, "tests"), (".py",)):
        if os.path.basename(path) == "test_layout.py":
            continue
        assert "/root/reference" not in open(path).read(), path


def test_required_layout():
    for rel in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/ts_hip.h", "oracle/__init__.py",
                "tests/golden/PROVENANCE.txt", "tools/gen_golden.py", "profiles", "temporalstereo_amd/csrc/block_cost.hip"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_oracle_headers_say_test_infrastructure():
    for path in _sources(os.path.join(ROOT, "oracle"), (".py",)):
        head = open(path).read(600).lower()
        assert "test infrastructure" in head, path


def test_bench_lines_fit_120_columns():
    """bench.py and its legs stay readable in a review pane."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in [os.path.join(root, "bench.py")] + sorted(glob.glob(os.path.join(root, "benchlegs", "*.py"))):
        with open(path) as f:
            long = [(n, len(line.rstrip("\n"))) for n, line in enumerate(f, 1) if len(line.rstrip("\n")) > 120]
        assert not long, "%s: lines over 120 columns: %s" % (os.path.relpath(path, root), long[:5])
