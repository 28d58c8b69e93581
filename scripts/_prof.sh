cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/g/bench_prof.json 2> gpurun_out/g/prof.err
find gpurun_out/g -name '*kernel_trace.csv' -delete
grep -E "attn_|layernorm" gpurun_out/g/prof/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-220
