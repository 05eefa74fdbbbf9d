"""In-tree native builds (no JIT cache: the .so files travel with the repo snapshot).

* ``libproxqp_hip.so``  -- the product: HIP kernels (gfx950) + the C-ABI, built with hipcc.
* ``libpqp_randqp.so``  -- host-only synthetic QP generator (g++).

The CPU oracle (``oracle/liboracle.so``) and the kernel-logic emulator
(``tests/emu/libpqp_emu.so``) are test infrastructure; they are built by
``__graft_entry__.build()`` / the test fixtures, never loaded from this package.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "proxsuite_amd" / "csrc"
INCLUDE = ROOT / "include"

HIP_LIB = CSRC / "libproxqp_hip.so"
RANDQP_LIB = CSRC / "libpqp_randqp.so"

HOST_CXXFLAGS = ["-O3", "-march=x86-64-v3", "-mtune=generic", "-std=gnu++17", "-fPIC", "-fopenmp"]


def _newer(target: Path, sources) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), r.stdout, r.stderr))
    return r


def build_randqp(force: bool = False) -> Path:
    src = [CSRC / "random_qp.cpp"]
    if not force and _newer(RANDQP_LIB, src):
        return RANDQP_LIB
    _run(["g++", *HOST_CXXFLAGS, "-shared", "-o", str(RANDQP_LIB), *map(str, src)])
    return RANDQP_LIB


def hip_sources():
    return [CSRC / "pqp_capi.hip"]


def hip_headers():
    return sorted(CSRC.glob("*.hpp")) + sorted(INCLUDE.glob("*.h"))


def build_hip(force: bool = False, extra_flags=()) -> Path:
    """Cross-compiles for gfx950 (works without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = hip_sources()
    if not force and _newer(HIP_LIB, list(srcs) + hip_headers()):
        return HIP_LIB
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=fast", "-I", str(INCLUDE), "-I", str(CSRC), *extra_flags,
           "-o", str(HIP_LIB), *map(str, srcs)]
    _run(cmd)
    return HIP_LIB


def build_oracle(force: bool = False) -> Path:
    odir = ROOT / "oracle"
    lib = odir / "liboracle.so"
    srcs = [odir / "proxqp_oracle.cpp", odir / "oracle_capi.cpp", odir / "proxqp_oracle.hpp",
            odir / "ldlt_oracle.hpp", INCLUDE / "pqp_types.h", odir / "Makefile"]
    if not force and _newer(lib, srcs):
        return lib
    if force and lib.exists():
        os.remove(lib)
    _run(["make", "-C", str(odir), "-B" if force else "-s", "liboracle.so"])
    return lib
