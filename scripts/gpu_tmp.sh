export TMPDIR=/tmp
NS=build/variants/libproxqp_hip_nostage.so
P=proxsuite_amd/csrc/libproxqp_hip.so
for off in 0 1; do echo "PQP_STAGE_OFF=$off"; for B in 1 128; do PQP_STAGE_OFF=$off timeout 120 python scripts/gpu_time_shape.py $B 100 50 100 0 0 3 $NS $P 2>&1 | grep -v "amdgpu.ids\|bit-id"; done;  PQP_STAGE_OFF=$off timeout 120 python scripts/gpu_time_libs.py c1 3 $NS $P 2>&1 | grep -v amdgpu.ids; done
