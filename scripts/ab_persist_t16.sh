mkdir -p gpurun_out/r03_aa
for m in 0 1 0 1; do VF_GEMM_PERSIST_T16=$m python bench.py --no-cpu-baseline --no-cfg3 --no-pipelined 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist_t16=$m', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['batch_of_8']['value'])"; done > gpurun_out/r03_aa/bench_persist_t16_ab.log 2>&1
cat gpurun_out/r03_aa/bench_persist_t16_ab.log
python scripts/shape_breakdown.py 32 2>/dev/null | grep -v amdgpu > gpurun_out/r03_aa/shape_breakdown.txt; head -14 gpurun_out/r03_aa/shape_breakdown.txt
VF_GEMM_PERSIST_T16=1 python scripts/shape_breakdown.py 32 2>/dev/null | grep "t16" 
