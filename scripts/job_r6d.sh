cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6d
python scripts/probes/concurrency_probe.py > gpurun_out/r6d/concurrency_probe.log 2>&1; tail -30 gpurun_out/r6d/concurrency_probe.log
PYTORCH_NO_CUDA_MEMORY_CACHING=1 python scripts/probes/overlap_diag.py 24 > gpurun_out/r6d/overlap_diag_nocache.log 2>&1; tail -5 gpurun_out/r6d/overlap_diag_nocache.log
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "one_block_per_window" > gpurun_out/r6d/t_dh32.log 2>&1; tail -3 gpurun_out/r6d/t_dh32.log
python scripts/vcf2exp_e2e.py --genes 512 --workers 1 --batch 32 --cores 2 --passes 3 2>&1 | grep "^pass" > gpurun_out/r6d/e2e.log
python scripts/vcf2exp_e2e.py --genes 512 --workers 1 --batch 32 --cores 1 --passes 3 2>&1 | grep "^pass" >> gpurun_out/r6d/e2e.log
cat gpurun_out/r6d/e2e.log
