#!/bin/bash
# The device sources on the CPU emulator under AddressSanitizer (the GPU pool offers no sanitizer: CPU build only).
#   scripts/dev/emu_asan.sh <wave|workgroup> <n,n_eq,n_in,B> [...more shape lists]
# Round 6: found the emulated range check of the one-wavefront kernel's buffer loads reading in front of a matrix for the
# items past the end of a pass (extent 0, stale row offset) -- the hardware fetches nothing there; fixed in dw_load_row.
R=$(cd $(dirname $0)/../.. && pwd)
csrc=$R/proxsuite_amd/csrc
LIB=/tmp/libpqp_emu_asan.so
if [ ! -f $LIB ] || [ $csrc/pqp_dwave.hpp -nt $LIB ] || [ $csrc/pqp_solver.hpp -nt $LIB ] || [ $csrc/pqp_block.hpp -nt $LIB ]; then
  g++ -std=gnu++17 -fPIC -shared -O1 -g -fsanitize=address -fno-omit-frame-pointer -pthread -fno-strict-aliasing -DPQP_STATS \
      -Wno-unknown-pragmas -Wno-attributes -I $R/tests/emu/include -I $R/include -I $csrc \
      -x c++ $csrc/pqp_capi.hip $csrc/pqp_multi.hip $csrc/pqp_kernels.hip $csrc/pqp_calib.hip $R/tests/emu/hip_emu.cpp -o $LIB || exit 1
fi
K=$1; shift
for sh in "$@"; do
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python - $K $sh <<'PY'
import os, sys
root = os.environ.get("PQP_ROOT") or os.getcwd()
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "emu"))
os.environ["PQP_DENSE_KERNEL"] = sys.argv[1]
import parity_cases as pc
from proxsuite_amd import _native as N
from proxsuite_amd.utils import random_qp as R
from oracle import oracle as O
lib = N.NativeLib("/tmp/libpqp_emu_asan.so")
n, ne, ni, B = (int(v) for v in sys.argv[2].split(","))
pc.case_random_batch(lib, O, R, n, ne, ni, B)
print("asan clean:", sys.argv[1], (n, ne, ni, B))
PY
done
