"""Seconds each test of the reference's test/src/dense_qp_wrapper.py takes against the drop-in on the CPU emulator
-> tests/golden/reference_suite_times.json (tests/test_reference_python_suite.py picks its default subset from it).
Runs here only (reads /root/reference; nothing is copied).  ~50 min: three tests of the file solve hundreds of QPs."""
import contextlib
import importlib.util
import io
import json
import os
import sys
import time
import unittest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build as emu_build  # noqa: E402
from proxsuite_amd import _native as N  # noqa: E402

N._lib = N.NativeLib(emu_build.build())
out = {}
for name in sys.argv[1:] or ["dense_qp_wrapper"]:
    spec = importlib.util.spec_from_file_location("reference_" + name, "/root/reference/test/src/%s.py" % name)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def walk(s):
        for t in s:
            if isinstance(t, unittest.TestSuite):
                yield from walk(t)
            else:
                yield t
    out[name] = {}
    for t in walk(unittest.defaultTestLoader.loadTestsFromModule(mod)):
        r = unittest.TestResult()
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            t.run(r)
        assert r.wasSuccessful(), (t.id(), r.failures + r.errors)
        out[name][t.id().split(".")[-1]] = round(time.time() - t0, 2)
        print(t.id().split(".")[-1], out[name][t.id().split(".")[-1]], flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reference_suite_times.json"), "w"), indent=1, sort_keys=True)
