set -x
mkdir -p gpurun_out/i
python -m pytest tests -m gpu -x -q > gpurun_out/i/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/i/pytest_gpu.log
python bench.py > gpurun_out/i/bench.json 2> gpurun_out/i/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/i/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/i/bench_prof.json 2> gpurun_out/i/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/i/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/i/pmc_fetch.json 2> gpurun_out/i/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/i/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/i/pmc_write.json 2> gpurun_out/i/pmc_write.err
ls -la gpurun_out/i/prof/*/ | head; du -sh gpurun_out/i
find gpurun_out/i -name '*kernel_trace.csv' -size +20M -delete
find gpurun_out/i/pmc_fetch gpurun_out/i/pmc_write -name '*counter_collection.csv' | head
tail -3 gpurun_out/i/pytest_gpu.log; cat gpurun_out/i/bench.json
python scripts/pmc_summary.py gpurun_out/i/pmc_hbm_traffic.json i/pmc > /dev/null
find gpurun_out/i -name '*counter_collection.csv' -delete
find gpurun_out/i -name '*kernel_trace.csv' -delete
du -sh gpurun_out/i
