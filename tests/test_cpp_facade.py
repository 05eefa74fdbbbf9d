"""The C++17 facade (include/proxsuite/...) compiled with g++ and run as a program
(tests/cpp/facade_test.cpp).  CPU: linked against the SIMT-emulator build of the device code
(test-only).  GPU (`-m gpu`): linked against libproxqp_hip.so."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "cpp" / "facade_test.cpp"
CSRC = ROOT / "proxsuite_amd" / "csrc"


def _compile(out, libdir, libname, extra=()):
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(SRC), "-o",
           str(out), "-L", str(libdir), "-l" + libname, "-L", str(CSRC), "-lpqp_randqp",
           "-Wl,-rpath," + str(libdir), "-Wl,-rpath," + str(CSRC), "-pthread"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run(exe):
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failure(s)" in r.stdout


def test_facade_on_emulator(randqp, tmp_path):
    sys.path.insert(0, str(ROOT / "tests" / "emu"))
    import build as emu_build
    lib = Path(emu_build.build())
    _run(_compile(tmp_path / "facade_emu", lib.parent, "pqp_emu"))


@pytest.mark.gpu
def test_facade_on_gpu(randqp, tmp_path):
    assert (CSRC / "libproxqp_hip.so").exists(), "build libproxqp_hip.so first (__graft_entry__.build())"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    _run(_compile(tmp_path / "facade_hip", CSRC, "proxqp_hip",
                  extra=["-L", rocm + "/lib", "-Wl,-rpath-link," + rocm + "/lib", "-Wl,-rpath," + rocm + "/lib"]))
