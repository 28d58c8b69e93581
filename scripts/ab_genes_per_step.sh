mkdir -p gpurun_out/r03_w
for g in ${GPS:-8 12 16 8}; do python bench.py --steps 4 --warmup 2 --genes-per-step $g --no-cpu-baseline --no-cfg3 --no-pipelined 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('genes_per_step=$g', d['value'], d['ms_per_step'], d['roofline']['achieved'])"; done > gpurun_out/r03_w/gps.log 2>&1
cat gpurun_out/r03_w/gps.log
