mkdir -p gpurun_out/r03_ag
for l in libvf_hip_prev.so libvf_hip.so libvf_hip_prev.so libvf_hip.so; do python scripts/ab_lib.py $l 32 8 2>/dev/null | tail -1; done > gpurun_out/r03_ag/step_ab.log 2>&1; cat gpurun_out/r03_ag/step_ab.log
for l in libvf_hip_prev.so libvf_hip.so libvf_hip_prev.so libvf_hip.so; do VF_LIB=$l python scripts/producer_stagger_bench.py 2>/dev/null | grep -v amdgpu | sed "s/\[stagger=0\]/[$l]/"; done > gpurun_out/r03_ag/producer_ab.log 2>&1; cat gpurun_out/r03_ag/producer_ab.log
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_persist_r16_gpu.py -x -q -m gpu -k "gemm or persist" > gpurun_out/r03_ag/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03_ag/pytest.log; tail -3 gpurun_out/r03_ag/pytest.log
