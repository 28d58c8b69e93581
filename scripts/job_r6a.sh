cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6a
python scripts/probes/g8x_store_probe.py > gpurun_out/r6a/g8x_store_probe.log 2>&1; cat gpurun_out/r6a/g8x_store_probe.log
bash scripts/run_profile_set.sh r6a
