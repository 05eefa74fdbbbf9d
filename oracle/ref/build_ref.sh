#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/ref_batchqp: the REAL ProxSuite dense backend
# (headers under /root/reference/include, compiled where they lie -- nothing is copied) behind the
# small driver oracle/ref/ref_batchqp.cpp.  ProxSuite is header-only on top of Eigen3, which
# /root/reference does not vendor: the recipe looks for Eigen/Core and, when it is absent, says so
# and exits 0 (the oracle then stays "parity unpinned" against the binary; see oracle/README.md).
#   oracle/ref/build_ref.sh [--require]      (--require: exit 1 when the binary cannot be built)
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
REF=${PROXSUITE_REFERENCE:-/root/reference}
REQ=0; [ "${1:-}" = "--require" ] && REQ=1
if [ ! -d "$REF/include/proxsuite" ]; then
  echo "build_ref: no reference tree at $REF (the GPU box uses what was prebuilt): nothing to do"; exit $REQ
fi
EIGEN=""
for d in ${EIGEN3_INCLUDE_DIR:-} /usr/include/eigen3 /usr/local/include/eigen3 /opt/eigen3 "$REF/external/eigen" $(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])" 2>/dev/null)/eigen3; do
  [ -n "$d" ] && [ -f "$d/Eigen/Core" ] && EIGEN=$d && break
done
if [ -z "$EIGEN" ]; then
  echo "build_ref: Eigen3 not found (looked for Eigen/Core in \$EIGEN3_INCLUDE_DIR, /usr/include/eigen3, /usr/local/include/eigen3, /opt/eigen3, $REF/external/eigen): the reference binary is NOT built; parity stays pinned on the reference's fixtures and known answers only"
  exit $REQ
fi
mkdir -p "$OUT"
set -x
g++ -O3 -march=x86-64-v3 -std=gnu++17 -fopenmp -DNDEBUG -I "$HERE/stub" -I "$REF/include" -I "$EIGEN" \
    "$HERE/ref_batchqp.cpp" -o "$OUT/ref_batchqp" || exit 1
set +x
echo "build_ref: built $OUT/ref_batchqp against Eigen at $EIGEN"
