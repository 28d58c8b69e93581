# the default bench line's timed region, repeated (fresh process each): per-step times and allocator calls
mkdir -p gpurun_out/r03_y
for i in ${RUNS:-1 2 3 4 5 6}; do python bench.py --no-cpu-baseline --no-cfg3 --no-pipelined --no-kernel-timing ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i', d['value'], d['ms_per_step'], d['ms_of_each_timed_step_rank0'], d['device_allocations_inside_timed_region'])"; done > gpurun_out/r03_y/bench_repeat.log 2>&1
cat gpurun_out/r03_y/bench_repeat.log
