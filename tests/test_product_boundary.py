"""`-m "not gpu"`: the product path never touches the oracle, and fails loudly without its HIP library.

* nothing under proxsuite_amd/, proxsuite/, include/ imports, includes, links or names anything under oracle/;
  bench.py does so only in its cpu_baseline / algorithm-count legs and __graft_entry__ only in build() / smoke();
* on a box with a GPU the Python binding refuses to run when libproxqp_hip.so is missing (no CPU fallback exists
  to fall back to): checked here on the loader itself."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(*dirs, ext=(".py", ".hpp", ".h", ".hip", ".cpp")):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if "__pycache__" in base:
                continue
            for f in files:
                if f.endswith(ext):
                    yield os.path.join(base, f)


def test_product_sources_never_reach_the_oracle():
    pat = re.compile(r"(from\s+oracle|import\s+oracle|oracle/|liboracle|proxqp_oracle|oracle\.)")
    hits = []
    for path in _sources("proxsuite_amd", "proxsuite", "include"):
        for k, line in enumerate(open(path, errors="replace"), 1):
            if pat.search(line) and "test infrastructure" not in line.lower():
                code = line.split("#")[0] if path.endswith(".py") else line.split("//")[0]
                if code.strip().startswith(("*", "/*")):  # (inside a C block comment)
                    continue
                if pat.search(code):  # (comments may NAME the oracle, e.g. where a builder of test infrastructure lives)
                    hits.append("%s:%d: %s" % (os.path.relpath(path, ROOT), k, line.strip()))
    # proxsuite_amd/_build.py builds oracle/liboracle.so for the tests (build_oracle): building the checker is not using it
    hits = [h for h in hits if not h.startswith("proxsuite_amd/_build.py")]
    assert hits == [], "\n".join(hits)


def test_bench_uses_the_oracle_only_as_baseline_and_checker():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)(from oracle|import oracle)", src, re.M):
        assert len(m.group(1)) > 0, "bench.py imports the oracle at module level"  # only inside the cpu_baseline / check legs
    assert "oracle" in src  # (the cpu_baseline leg exists)


def test_missing_library_is_an_error_not_a_fallback(tmp_path, monkeypatch):
    from proxsuite_amd import _native as N
    with pytest.raises((OSError, RuntimeError, FileNotFoundError)):
        N.NativeLib(str(tmp_path / "libproxqp_hip.so"))
    monkeypatch.setenv("PQP_HIP_LIBRARY", str(tmp_path / "missing.so"))
    saved = N._lib
    N._lib = None
    try:
        with pytest.raises((OSError, RuntimeError, FileNotFoundError)):
            N.load()
    finally:
        N._lib = saved
