"""TEST INFRASTRUCTURE ONLY.

Builds tests/emu/libpqp_emu.so: the *unmodified* HIP sources of the product
(proxsuite_amd/csrc/pqp_capi.hip, pqp_kernels.hip + headers) compiled with g++ against the fiber-based
SIMT emulator (hip_emu.hpp/.cpp), exposing the same C-ABI as libproxqp_hip.so.
It exists so that kernel logic can be validated against the oracle on a box without a
GPU.  The product package never loads it (proxsuite_amd._native only ever opens
proxsuite_amd/csrc/libproxqp_hip.so).
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
LIB = HERE / "libpqp_emu.so"


def build_variant(tag, defines):
    """A second emulator library with extra -D switches (A/B of a code path on the CPU: tests/test_emu_parity.py)."""
    csrc = ROOT / "proxsuite_amd" / "csrc"
    out = HERE / ("libpqp_emu_%s.so" % tag)
    srcs = [csrc / "pqp_capi.hip", csrc / "pqp_multi.hip", csrc / "pqp_kernels.hip", csrc / "pqp_calib.hip", HERE / "hip_emu.cpp"]
    deps = srcs + [csrc / "pqp_block.hpp", csrc / "pqp_solver.hpp", csrc / "pqp_host.hpp", csrc / "pqp_diag.hpp", csrc / "pqp_dwave.hpp", HERE / "hip_emu.hpp", Path(__file__)]
    if out.exists() and all(d.stat().st_mtime <= out.stat().st_mtime for d in deps):
        return out
    cmd = ["g++", "-std=gnu++17", "-fPIC", "-shared", "-O2", "-pthread", "-fno-strict-aliasing", "-DPQP_STATS",
           *["-D" + d for d in defines], "-Wno-unknown-pragmas", "-Wno-attributes",
           "-I", str(HERE / "include"), "-I", str(ROOT / "include"), "-I", str(csrc),
           "-x", "c++", *map(str, srcs), "-o", str(out) + ".tmp%d" % os.getpid()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulator variant build failed:\n" + r.stdout + r.stderr)
    os.replace(str(out) + ".tmp%d" % os.getpid(), out)  # (atomic: parallel test workers may build at the same time)
    return out


def build(force=False, debug=False):
    # PQP_EMU_LIBRARY=<path>: the emulator tests load that build of the emulator library instead (scripts/dev/emu_asan.sh: the
    # AddressSanitizer build, with LD_PRELOAD=libasan.so in front of python)
    if os.environ.get("PQP_EMU_LIBRARY"):
        return Path(os.environ["PQP_EMU_LIBRARY"])
    csrc = ROOT / "proxsuite_amd" / "csrc"
    srcs = [csrc / "pqp_capi.hip", csrc / "pqp_multi.hip", csrc / "pqp_kernels.hip", csrc / "pqp_calib.hip", HERE / "hip_emu.cpp"]
    deps = srcs + [csrc / "pqp_block.hpp", csrc / "pqp_solver.hpp", csrc / "pqp_host.hpp", csrc / "pqp_diag.hpp", csrc / "pqp_dwave.hpp", HERE / "hip_emu.hpp",
                   HERE / "include" / "hip" / "hip_runtime.h", ROOT / "include" / "proxqp_hip.h",
                   ROOT / "include" / "pqp_types.h", Path(__file__)]
    if not force and LIB.exists() and all(d.stat().st_mtime <= LIB.stat().st_mtime for d in deps):
        return LIB
    opt = ["-O0", "-g"] if debug else ["-O2"]
    cmd = ["g++", "-std=gnu++17", "-fPIC", "-shared", *opt, "-pthread", "-fno-strict-aliasing", "-DPQP_STATS",
           "-Wno-unknown-pragmas", "-Wno-attributes",
           "-I", str(HERE / "include"), "-I", str(ROOT / "include"), "-I", str(csrc),
           "-x", "c++", *map(str, srcs), "-o", str(LIB) + ".tmp%d" % os.getpid()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + r.stdout + r.stderr)
    os.replace(str(LIB) + ".tmp%d" % os.getpid(), LIB)  # (atomic: parallel test workers may build at the same time)
    return LIB


if __name__ == "__main__":
    print(build(force=True, debug="--debug" in sys.argv))
