set -x
mkdir -p gpurun_out/f
python -m pytest tests -m gpu -x -q > gpurun_out/f/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/f/pytest_gpu.log
python bench.py > gpurun_out/f/bench.json 2> gpurun_out/f/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/f/bench_prof.json 2> gpurun_out/f/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/f/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/f/pmc_fetch.json 2> gpurun_out/f/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/f/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/f/pmc_write.json 2> gpurun_out/f/pmc_write.err
ls -la gpurun_out/f/prof/*/ | head; du -sh gpurun_out/f
find gpurun_out/f -name '*kernel_trace.csv' -size +20M -delete
find gpurun_out/f/pmc_fetch gpurun_out/f/pmc_write -name '*counter_collection.csv' | head
tail -3 gpurun_out/f/pytest_gpu.log; cat gpurun_out/f/bench.json
python scripts/pmc_summary.py gpurun_out/f/pmc_hbm_traffic.json f/pmc > /dev/null
find gpurun_out/f -name '*counter_collection.csv' -delete
find gpurun_out/f -name '*kernel_trace.csv' -delete
du -sh gpurun_out/f
