"""`-m "not gpu"`: the reference's OWN Python test files, UNMODIFIED, run against the drop-in.

`import proxsuite` resolves to this repository's package (proxsuite/ -> proxsuite_amd), whose dense backend is
driven here through the CPU SIMT emulator build of the device code (tests/emu -- TEST ONLY).  The files are loaded
from /root/reference/test/src at run time and never copied: the tests skip where the reference tree is absent (the
GPU box).  What passes: every dense test of dense_qp_solve.py (9), dense_qp_wrapper.py (46) and parallel_qp_solve.py
(2); the only test that cannot run is the sparse-backend one of parallel_qp_solve.py (out of scope).
The reference's Python EXAMPLES (examples/python/*.py) run the same way: all 19 that use the dense backend.
dense_qp_wrapper.py takes 49 min on the emulator (one fiber per GPU thread; two of its tests take 47 of them), so the
default run takes the 43 tests below its time budget and `PQP_REFERENCE_SUITE_FULL=1` runs all 46."""
import contextlib
import importlib.util
import io
import json
import os
import sys
import unittest

import pytest

from proxsuite_amd import _native as N

REF = "/root/reference/test/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")

# seconds of every test of dense_qp_wrapper.py on the emulator (tests/golden/reference_suite_times.json, written by
# scripts/time_reference_suite.py); the default run takes the ones under the budget
BUDGET_S = 6.0


@pytest.fixture(scope="module")
def emulated():
    import build as emu_build
    saved = N._lib
    N._lib = N.NativeLib(emu_build.build())
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    yield
    N._lib = saved


def _load(name):
    spec = importlib.util.spec_from_file_location("reference_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)

    def walk(s):
        for t in s:
            if isinstance(t, unittest.TestSuite):
                yield from walk(t)
            else:
                yield t
    return {t.id().split(".")[-1]: t for t in walk(unittest.defaultTestLoader.loadTestsFromModule(mod))}


def _run(tests):
    bad = []
    for name, t in tests.items():
        r = unittest.TestResult()
        with contextlib.redirect_stdout(io.StringIO()):
            t.run(r)
        if not r.wasSuccessful():
            bad.append((name, (r.failures + r.errors)[0][1].strip().splitlines()[-1]))
    return bad


def test_dense_qp_solve_py(emulated):
    """test/src/dense_qp_solve.py: proxsuite.proxqp.dense.solve(...) -- all 9 tests"""
    tests = _load("dense_qp_solve")
    assert len(tests) == 9
    assert _run(tests) == []


def test_parallel_qp_solve_py(emulated):
    """test/src/parallel_qp_solve.py: solve_in_parallel on a list of QPs and on a BatchQP (the sparse one is out of scope)"""
    tests = {k: t for k, t in _load("parallel_qp_solve").items() if "sparse" not in k}
    assert len(tests) == 2
    assert _run(tests) == []


def test_dense_qp_wrapper_py(emulated):
    """test/src/dense_qp_wrapper.py: the QP object -- init / solve / update / warm starts / settings / boxes ..."""
    tests = _load("dense_qp_wrapper")
    assert len(tests) == 46
    if os.environ.get("PQP_REFERENCE_SUITE_FULL") != "1":
        times = json.load(open(os.path.join(HERE, "golden", "reference_suite_times.json")))["dense_qp_wrapper"]
        assert set(times) == set(tests), "the reference's test list changed: re-run scripts/time_reference_suite.py"
        tests = {k: t for k, t in tests.items() if times[k] <= BUDGET_S}
        assert len(tests) >= 40
    assert _run(tests) == []


def test_examples_py(emulated):
    """examples/python/*.py of the reference, as scripts: every one that does not use the sparse backend (4) or
    cvxpy (qplayer_sudoku.py; not installed here)"""
    import runpy
    ex = os.path.join(os.path.dirname(os.path.dirname(REF)), "examples", "python")
    ran, bad = 0, []
    cwd = os.getcwd()
    sys.path.insert(0, ex)  # (the examples import their util.py)
    os.chdir(ex)            # (and load data/ by relative path)
    try:
        for f in sorted(os.listdir(ex)):
            if not f.endswith(".py") or f == "util.py":
                continue
            src = open(os.path.join(ex, f)).read()
            if "proxqp.sparse" in src or "cvxpy" in src:
                continue
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    runpy.run_path(os.path.join(ex, f), run_name="__main__")
                ran += 1
            except BaseException as e:  # noqa: BLE001 (a script may call exit())
                bad.append((f, type(e).__name__, str(e)[:200]))
    finally:
        os.chdir(cwd)
        sys.path.remove(ex)
    assert bad == [] and ran >= 17, (ran, bad)


def test_benchmark_timings_parallel_py(emulated):
    """benchmark/timings-parallel.py of the reference -- BatchQP / VectorQP set-up, `solve_in_parallel` over thread
    counts, `qp.solve()` serially, and `dense.solve_no_gil` serially and under a ThreadPoolExecutor (:102-137) -- run
    from the reference tree with only its two workload constants made emulator-sized (`problem_specs`, `num_qps`: an AST
    rewrite of those two assignments at load time; every call the script makes is the reference's)."""
    import ast
    path = os.path.join(os.path.dirname(os.path.dirname(REF)), "benchmark", "timings-parallel.py")
    tree = ast.parse(open(path).read(), filename=path)
    seen = set()
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id == "problem_specs":
                node.value = ast.parse("[(10, 4, 4), (16, 6, 6)]", mode="eval").body
                seen.add("problem_specs")
            elif node.targets[0].id == "num_qps":
                node.value = ast.Constant(6)
                seen.add("num_qps")
    assert seen == {"problem_specs", "num_qps"}
    ast.fix_missing_locations(tree)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        exec(compile(tree, path, "exec"), {"__name__": "__main__"})
    text = out.getvalue()
    for key in ("setup_batch_of_qps", "solve_in_parallel_heuristics_threads", "qp_solve_serial", "solve_fun_serial", "solve_fun_parallel"):
        assert key in text, (key, text[-400:])
