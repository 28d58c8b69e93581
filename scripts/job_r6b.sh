cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6b
python scripts/probes/overlap_diag.py 64 > gpurun_out/r6b/overlap_diag.log 2>&1; tail -6 gpurun_out/r6b/overlap_diag.log
python scripts/probes/g8x_store_probe.py > gpurun_out/r6b/g8x_store_probe.log 2>&1; cat gpurun_out/r6b/g8x_store_probe.log
