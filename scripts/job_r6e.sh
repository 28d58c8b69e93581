cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6e
python scripts/probes/concurrency_probe2.py > gpurun_out/r6e/concurrency_probe2.log 2>&1; grep -v "amdgpu\|^gene\|^total" gpurun_out/r6e/concurrency_probe2.log | tail -60
