# fp16 trunk copy (VF_TRUNK16=f16, default) against the fp32 trunk (VF_TRUNK16=0): op test, model parity subset, bench A/B
mkdir -p gpurun_out/r03_aa
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fp16_trunk or gemm_ln_producer" > gpurun_out/r03_aa/pytest_ops.log 2>&1; echo "rc=$?" >> gpurun_out/r03_aa/pytest_ops.log; tail -4 gpurun_out/r03_aa/pytest_ops.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r03_aa/pytest_model.log 2>&1; echo "rc=$?" >> gpurun_out/r03_aa/pytest_model.log; tail -15 gpurun_out/r03_aa/pytest_model.log
for m in 0 f16 0 f16; do VF_TRUNK16=$m python bench.py --no-cpu-baseline --no-cfg3 --no-pipelined 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trunk16=$m', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['batch_of_8']['value'])"; done > gpurun_out/r03_aa/bench_ab.log 2>&1
cat gpurun_out/r03_aa/bench_ab.log
