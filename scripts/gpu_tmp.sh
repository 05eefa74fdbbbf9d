export TMPDIR=/tmp
for l in build/variants/libproxqp_hip_diagw1.so build/variants/libproxqp_hip_diagw3.so; do
  echo "== $l"; B=512 timeout 60 python scripts/gpu_time_libs.py c5 1 $l 2>&1 | grep -v amdgpu.ids | tail -3; echo "rc=$?"
done
echo "== both product+w1"; B=512 timeout 90 python scripts/gpu_time_libs.py c5 1 proxsuite_amd/csrc/libproxqp_hip.so build/variants/libproxqp_hip_diagw1.so 2>&1 | grep -v amdgpu.ids | tail -3
