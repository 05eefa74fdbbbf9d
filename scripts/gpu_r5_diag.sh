#!/bin/bash
# the one-wavefront diagonal kernel: parity against the oracle on C5 / C5box, and its time at 1 / 2 / 3 waves per SIMD
export TMPDIR=/tmp
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
python scripts/gpu_box_probe.py c2 c5 2>&1 | grep BOX
for bx in 0 1; do timeout 600 python scripts/gpu_c5_check.py 200 ${NCHK:-1024} $bx 2>&1 | tail -4; done | tee $O/check.txt
for w in c5 c5box; do
  timeout 600 python scripts/gpu_time_libs.py $w 3 proxsuite_amd/csrc/libproxqp_hip.so build/variants/libproxqp_hip_diagw1.so build/variants/libproxqp_hip_diagw3.so 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt
done
PQP_DIAG_KERNEL=workgroup timeout 600 python scripts/gpu_time_libs.py c5 2 proxsuite_amd/csrc/libproxqp_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt
