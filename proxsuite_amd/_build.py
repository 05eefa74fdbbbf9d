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
# the same library with the per-phase device timers and event counters compiled in (-DPQP_STATS):
# used by bench.py OUTSIDE its timed region and by the profiling scripts, never by the product path
HIP_STATS_LIB = CSRC / "libproxqp_hip_stats.so"
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


# pqp_kernels.hip is compiled once per kernel family (see its header): every solve kernel is
# ~350 KB of inlined code and takes about a minute of hipcc time, so the objects are built in
# parallel and linked into one shared library.
KERNEL_TUS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14, 15, 16, 17, 18)
OBJ_DIR = ROOT / "build" / "obj"


def hip_sources():
    return [CSRC / "pqp_capi.hip", CSRC / "pqp_multi.hip", CSRC / "pqp_kernels.hip", CSRC / "pqp_calib.hip"]


def hip_headers():
    return sorted(CSRC.glob("*.hpp")) + sorted(INCLUDE.glob("*.h"))


# Register allocation of the one-kernel solver is the lever: sinking
# instructions into the loops that use them and NOT hoisting loop invariants out of them cuts the
# VGPR spills of the C2 kernel from 517 to 163 (+9 % QPs/s, profiles/r02_ab_compiler_flags.txt); without
# loop strength reduction the mat-vec loops recompute their addresses from one index instead of
# carrying a pointer per load in flight (128-VGPR kernel: 319 -> 206 spilled VGPRs, +4.5 % QPs/s).
CODEGEN_FLAGS = ["-mllvm", "-sink-insts-to-avoid-spills", "-mllvm", "-disable-machine-licm"]
# per translation unit (kernel family, csrc/pqp_kernels.hip): measured on the workload each one serves --
# C2 +4.3 %, C1 +3.4 %, C4 +2 % with -disable-lsr; the structured / boxed 256-thread kernel (C5: -17 %) and the
# 512-thread kernels (dense-backend shape: -7 %) keep loop strength reduction
TU_FLAGS = {17: ["-mllvm", "-disable-lsr"],  # the one-wavefront dense kernel: 7.65 -> 7.37 ms per 2048 C2 QPs (profiles/r06_ab_dwave.txt section 13)
            18: ["-mllvm", "-disable-lsr", "-DPQP_ZG_DEPTH=4"],  # the prologue kernel: 4 k-steps of operand loads in flight per lane in its Z / G build (8 elsewhere): 1.026 -> 0.985 ms per 2048 C2 QPs; 2 / 3 / 5 / 6 / 12: 1.000 / 1.003 / 0.986 / 1.020 / 1.467 ms
            1: ["-mllvm", "-disable-lsr"], 4: ["-mllvm", "-disable-lsr"], 7: ["-mllvm", "-disable-lsr"], 13: ["-mllvm", "-disable-lsr"], 15: ["-mllvm", "-disable-lsr"]}


def hip_flags(extra_flags=()):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", *CODEGEN_FLAGS,
            "-Rpass-analysis=kernel-resource-usage",  # per-kernel registers / spills / scratch, parsed below
            "-I", str(INCLUDE), "-I", str(CSRC), *extra_flags]


def parse_kernel_resources(stderr: str) -> dict:
    """The AMDGPU backend's kernel-resource-usage remarks -> {demangled-ish kernel name: {field: value}}.
    The register allocation of the one-kernel solver decides its speed, and three
    internal -mllvm switches shape it: every build records what it got (build/kernel_resources.json,
    copied to profiles/ by __graft_entry__.build) so that a toolchain change cannot shift it silently."""
    import re
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None:
            key = re.sub(r"\s*\[.*\]", "", key).strip().replace(" ", "_")
            try:
                cur[key] = int(val)
            except ValueError:
                cur[key] = val
    return out


def kernel_label(mangled: str) -> str:
    """_Z16pqp_solve_kernelILi256ELi4ELi1EEv... -> pqp_solve_kernel<256,4,1>"""
    import re
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I((?:Li\d+E)+)E", mangled)
    if not m:
        return mangled
    return "%s<%s>" % (m.group(1), ",".join(re.findall(r"Li(\d+)E", m.group(2))))


def flags_tag(extra_flags) -> str:
    """stable name of a flag set (object directory / object suffix of a variant build)"""
    import hashlib
    return "v" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]


def build_hip(force: bool = False, extra_flags=(), out: Path = None, tus=KERNEL_TUS) -> Path:
    """Cross-compiles for gfx950 (works without a GPU).  `out` / `extra_flags` build a variant
    of the library somewhere else (A/B runs: scripts/gpu_ab.sh); `tus` restricts the kernel
    families that are compiled (development only: the launchers of the others are then missing
    and the link fails unless the variant is never asked for them... so keep the default)."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = Path(out) if out else HIP_LIB
    deps = list(hip_sources()) + hip_headers() + [Path(__file__)]
    if not force and _newer(lib, deps):
        return lib
    tag = "default" if not extra_flags else flags_tag(extra_flags)
    odir = OBJ_DIR / tag
    odir.mkdir(parents=True, exist_ok=True)
    flags = hip_flags(extra_flags)
    jobs = [([hipcc, *flags, "-c", str(CSRC / "pqp_capi.hip"), "-o", str(odir / "capi.o")], odir / "capi.o"),
            ([hipcc, *flags, "-c", str(CSRC / "pqp_multi.hip"), "-o", str(odir / "multi.o")], odir / "multi.o"),
            ([hipcc, *flags, "-c", str(CSRC / "pqp_calib.hip"), "-o", str(odir / "calib.o")], odir / "calib.o")]
    for k in tus:
        o = odir / ("kernels_%d.o" % k)
        jobs.append(([hipcc, *flags, *TU_FLAGS.get(k, []), "-DPQP_TU=%d" % k, "-c", str(CSRC / "pqp_kernels.hip"),
                      "-o", str(o)], o))
    # an object is stale when ITS source, a header or this recipe is newer (an edit of pqp_multi.hip does not recompile
    # sixteen kernel families)
    common = hip_headers() + [Path(__file__)]
    own = {"capi.o": "pqp_capi.hip", "multi.o": "pqp_multi.hip", "calib.o": "pqp_calib.hip"}
    # (pqp_dwave.hpp is parsed by every kernel family and instantiated by family 17 alone: inline templates, no code elsewhere)
    only = {"pqp_dwave.hpp": ("kernels_17.o",)}
    def deps_of(o):
        return [h for h in common if h.name not in only or o.name in only[h.name]] + [CSRC / own.get(o.name, "pqp_kernels.hip")]
    todo = [j for j in jobs if force or not _newer(j[1], deps_of(j[1]))]
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        results = list(ex.map(lambda j: _run(j[0]), todo))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *[str(j[1]) for j in jobs]])
    # registers / spills / scratch of every kernel that was (re)compiled, merged into the record of this build tag
    import json
    rec_path = odir / "kernel_resources.json"
    rec = json.loads(rec_path.read_text()) if rec_path.exists() else {}
    for j, r in zip(todo, results):
        rec = {k: v for k, v in rec.items() if v.get("object") != j[1].name}  # (what that object held before this compile)
        for name, fields in parse_kernel_resources(r.stderr).items():
            rec[kernel_label(name)] = dict(fields, object=j[1].name)
    rec_path.write_text(json.dumps(rec, indent=1, sort_keys=True))
    return lib


def kernel_resources(tag: str = "default") -> dict:
    """what the last build of `tag` recorded (see parse_kernel_resources)"""
    import json
    p = OBJ_DIR / tag / "kernel_resources.json"
    rec = json.loads(p.read_text()) if p.exists() else {}
    # (the record accumulates over builds: entries of translation units that no longer exist are dropped)
    live = {"kernels_%d.o" % k for k in KERNEL_TUS}
    return {k: v for k, v in rec.items() if v.get("object") in live}


def build_hip_stats(force: bool = False) -> Path:
    lib = HIP_STATS_LIB
    deps = list(hip_sources()) + hip_headers() + [Path(__file__)]
    if not force and _newer(lib, deps):
        return lib
    return build_hip(extra_flags=("-DPQP_STATS",), out=lib)  # (stale objects only)


VARIANT_DIR = CSRC / "variants"


def build_hip_variants(force: bool = False):
    """Register-budget variants of the 512-thread latency kernel (PQP_WPS_512 = 3 and 4; the product
    uses 2), for the GPU regression test that sweeps them (tests/test_gpu_parity.py): round 1 saw
    NaNs at (512, 4) with a kernel that has since been rewritten; the sweep keeps watch.  Only the
    translation unit of that kernel is recompiled; the other objects are the product's."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    VARIANT_DIR.mkdir(exist_ok=True)
    out = []
    deps = list(hip_sources()) + hip_headers() + [Path(__file__)]
    build_hip()  # the product objects the variants link against
    base = OBJ_DIR / "default"
    for w in (3, 4):
        lib = VARIANT_DIR / ("libproxqp_hip_wps512_%d.so" % w)
        if force or not _newer(lib, deps):
            o3 = base / ("kernels_3_wps%d.o" % w)
            _run([hipcc, *hip_flags(("-DPQP_WPS_512=%d" % w,)), *TU_FLAGS.get(3, []), "-DPQP_TU=3", "-c",
                  str(CSRC / "pqp_kernels.hip"),
                  "-o", str(o3)])
            objs = [base / "capi.o", base / "multi.o", base / "calib.o"] + [o3 if k == 3 else base / ("kernels_%d.o" % k) for k in KERNEL_TUS]
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)])
        out.append(lib)
    return out


def build_tu_variant(tu, extra_flags, out: Path):
    """A/B partner that differs from the product in ONE kernel family: translation unit `tu` recompiled with
    `extra_flags`, every other object taken from the product build (seconds instead of minutes).  Returns (library path,
    {kernel: resources})."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    build_hip()
    base = OBJ_DIR / "default"
    out = Path(out)
    out.parent.mkdir(parents=True, exist_ok=True)
    tus = [tu] if isinstance(tu, int) else list(tu)
    repl, res = {}, {}
    for t in tus:
        o = base / ("kernels_%d_%s.o" % (t, flags_tag(extra_flags)))
        r = _run([hipcc, *hip_flags(tuple(extra_flags)), *TU_FLAGS.get(t, []), "-DPQP_TU=%d" % t, "-c",
                  str(CSRC / "pqp_kernels.hip"), "-o", str(o)])
        repl[t] = o
        res.update({kernel_label(k): v for k, v in parse_kernel_resources(r.stderr).items()})
    objs = [base / "capi.o", base / "multi.o", base / "calib.o"] + [repl.get(k, base / ("kernels_%d.o" % k)) for k in KERNEL_TUS]
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    return out, res


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


def kernel_sources_sha(diag: bool = False, dwave: bool = False) -> str:
    """sha256 (first 16 hex digits) of the device sources a solve kernel is compiled from: what a PMC traffic figure in
    profiles/pmc_traffic.json was measured on (scripts/merge_pmc_traffic.py stamps it, bench.py compares)"""
    import hashlib
    files = ["pqp_solver.hpp", "pqp_block.hpp"] + (["pqp_diag.hpp"] if diag or dwave else []) + (["pqp_dwave.hpp"] if dwave else [])
    import re
    h = hashlib.sha256()
    for f in files:
        # the CODE: comments and blank lines do not make a measurement stale
        text = (CSRC / f).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        lines = [re.sub(r"//.*$", "", ln).rstrip() for ln in text.splitlines()]
        h.update("\n".join(ln for ln in lines if ln.strip()).encode())
    return h.hexdigest()[:16]


def freeze_kernel_resources():
    """tests/golden/kernel_resources_expected.json <- the record of the current product build (run after a kernel
    change has been measured on the GPU and accepted: tests/test_kernel_resources.py then guards it)"""
    import json
    build_hip()
    rec = kernel_resources()
    keep = ("VGPRs", "AGPRs", "VGPRs_Spill", "SGPRs_Spill", "ScratchSize", "Occupancy")
    out = {k: {f: v[f] for f in keep if f in v} for k, v in sorted(rec.items()) if k.startswith("pqp_")}
    p = ROOT / "tests" / "golden" / "kernel_resources_expected.json"
    p.write_text(json.dumps(out, indent=1, sort_keys=True))
    return p


if __name__ == "__main__":
    import sys
    if "--freeze" in sys.argv:
        print(freeze_kernel_resources())
