#!/usr/bin/env python
"""Development: one kernel translation unit compiled from a source directory of its own (a copy of csrc/ with an edited or an
older header) and linked with the product's other objects into an A/B library -- seconds to a minute instead of the full build.
    python scripts/dev/build_tu.py <tu> <csrc dir> <out.so> [extra flags...]"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proxsuite_amd import _build as B

tus = [int(t) for t in sys.argv[1].split(",")]
src, out, extra = sys.argv[2], sys.argv[3], sys.argv[4:]
base = B.OBJ_DIR / os.environ.get("OBJ_TAG", "default")  # OBJ_TAG=v29fce601: the -DPQP_STATS objects
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
flags = [f for f in B.hip_flags(tuple(extra))]
flags[flags.index(str(B.CSRC))] = src  # -I <csrc> -> the variant's directory
for drop in filter(None, os.environ.get("DROP_FLAGS", "").split(",")):  # e.g. DROP_FLAGS=-sink-insts-to-avoid-spills
    i = flags.index(drop)
    del flags[i - 1:i + 1]  # ("-mllvm", flag)
repl = {}
for tu in tus:
    o = os.path.abspath(out) + ".k%d.o" % tu
    t0 = time.time()
    r = subprocess.run(["hipcc", *flags, *B.TU_FLAGS.get(tu, []), "-DPQP_TU=%d" % tu, "-c", os.path.join(src, "pqp_kernels.hip"), "-o", o],
                       capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-6000:]); sys.exit(1)
    for k, v in B.parse_kernel_resources(r.stderr).items():
        print(B.kernel_label(k), {f: v[f] for f in ("VGPRs", "VGPRs_Spill", "SGPRs_Spill", "ScratchSize", "Occupancy") if f in v})
    print("compiled TU %d in %.0f s" % (tu, time.time() - t0))
    repl[tu] = o
capi = base / "capi.o"
if os.environ.get("CAPI"):
    capi = os.path.abspath(out) + ".capi.o"
    r = subprocess.run(["hipcc", *flags, "-c", os.path.join(src, "pqp_capi.hip"), "-o", capi], capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-6000:]); sys.exit(1)
objs = [capi, base / "multi.o", base / "calib.o"] + [repl.get(k, base / ("kernels_%d.o" % k)) for k in B.KERNEL_TUS]
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *map(str, objs)], check=True)
print("linked", out)
