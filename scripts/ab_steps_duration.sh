# does the step time depend on how long the timed region runs (sustained clocks)?  genes-per-step x steps
mkdir -p gpurun_out/r03_y
for gs in "8 6" "8 24" "8 96" "32 2" "32 6" "32 24" "8 6" "32 6"; do set -- $gs
python bench.py --steps $2 --warmup 2 --genes-per-step $1 --no-cpu-baseline --no-cfg3 --no-pipelined --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('genes_per_step=$1 steps=$2', d['value'], d['ms_per_step'])"; done > gpurun_out/r03_y/steps_duration.log 2>&1
cat gpurun_out/r03_y/steps_duration.log
