cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6f
python scripts/probes/concurrency_probe3.py > gpurun_out/r6f/concurrency_probe3.log 2>&1; grep -v amdgpu gpurun_out/r6f/concurrency_probe3.log | tail -40
