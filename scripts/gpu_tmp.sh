export TMPDIR=/tmp
A=build/variants/libproxqp_hip_base.so
P=proxsuite_amd/csrc/libproxqp_hip.so
python scripts/gpu_box_probe.py c2 2>&1 | grep BOX
timeout 200 python scripts/gpu_time_libs.py c2 4 $A $P 2>&1 | grep -v amdgpu.ids
timeout 100 python scripts/gpu_bitcompare.py c2 $A $P 2>&1 | grep -v amdgpu.ids | tail -1
timeout 100 python scripts/gpu_time_libs.py c1 4 $A $P 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/gpu_time_libs.py c4 2 $A $P 2>&1 | grep -v amdgpu.ids
timeout 100 python scripts/gpu_bitcompare.py c4 $A $P 2>&1 | grep -v amdgpu.ids | tail -1
