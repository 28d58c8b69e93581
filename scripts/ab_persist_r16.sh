mkdir -p gpurun_out/r03_v
for m in 0 1 0 1; do VF_GEMM_PERSIST_R16=$m python scripts/producer_stagger_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/\[stagger=0\]/[persist_r16=$m]/" ; done > gpurun_out/r03_v/producer_ab.log 2>&1
timeout 1500 python -m pytest tests/test_persist_r16_gpu.py -x -q -m gpu > gpurun_out/r03_v/pytest_r16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_v/pytest_r16.log
for m in 0 1 0 1; do VF_GEMM_PERSIST_R16=$m python bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist_r16=$m', d['value'], d['ms_per_step'], d['roofline']['achieved'])" ; done > gpurun_out/r03_v/bench_ab.log 2>&1
cat gpurun_out/r03_v/producer_ab.log gpurun_out/r03_v/bench_ab.log; tail -5 gpurun_out/r03_v/pytest_r16.log
