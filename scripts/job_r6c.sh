cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6c
python scripts/probes/overlap_diag.py 48 > gpurun_out/r6c/overlap_diag.log 2>&1; tail -5 gpurun_out/r6c/overlap_diag.log
python scripts/probes/attn_probes_r6.py 32 > gpurun_out/r6c/attn_probes.log 2>&1; cat gpurun_out/r6c/attn_probes.log
python -m pytest tests -m gpu -x -q -s > gpurun_out/r6c/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6c/pytest_gpu.log
tail -3 gpurun_out/r6c/pytest_gpu.log; grep -a "\[headline\|\[trained-like" gpurun_out/r6c/pytest_gpu.log
python scripts/s2r_dims_sweep.py gpurun_out/r6c/s2r_dims.json > gpurun_out/r6c/s2r_dims.log 2>&1; tail -20 gpurun_out/r6c/s2r_dims.log
