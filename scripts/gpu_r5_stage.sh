#!/bin/bash
# LDS-staged matrices of small launches: streaming build (variant) against the product, interleaved on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
NS=build/variants/libproxqp_hip_nostage.so
P=proxsuite_amd/csrc/libproxqp_hip.so
python scripts/gpu_box_probe.py c2 c1 2>&1 | grep BOX
timeout 120 python scripts/gpu_time_libs.py c1 3 $NS $P 2>&1 | grep -v amdgpu.ids | tee $O/time.txt
timeout 120 python scripts/gpu_bitcompare.py c1 $NS $P 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/time.txt
for B in 1 64 256; do
  timeout 120 python scripts/gpu_time_shape.py $B 100 50 100 0 0 3 $NS $P 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt
done
timeout 120 python scripts/gpu_time_shape.py 128 50 25 50 1 0 3 $NS $P 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt
timeout 120 python scripts/gpu_time_shape.py 64 100 200 200 1 2 3 $NS $P 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt
