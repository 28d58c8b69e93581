#!/bin/bash
# round 6, final set part 2: the rocprofv3 passes again on W + K + K identical steps (--no-extra-rates), the two-thread test
mkdir -p gpurun_out/r6z
timeout 600 python -m pytest tests/test_concurrency_gpu.py -m gpu -q > gpurun_out/r6z/pytest_concurrency.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6z/pytest_concurrency.log
bash scripts/run_profile_set.sh r6z --prof-only
