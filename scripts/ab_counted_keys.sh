# A/B on one box: CRE layers' context cross attention over the 9 distinct label rows with counts (default) vs over the gathered rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "counted or row_map" > gpurun_out/r5e/t1.log 2>&1; tail -3 gpurun_out/r5e/t1.log
python -m pytest tests/test_model_gpu.py tests/test_ln_heal_gpu.py tests/test_trained_like_gpu.py -x -q -m gpu > gpurun_out/r5e/t2.log 2>&1; tail -3 gpurun_out/r5e/t2.log
for rep in 1 2; do
for flag in 1 0; do
python - > gpurun_out/r5e/bench_c${flag}_$rep.json 2> gpurun_out/r5e/bench_c${flag}_$rep.err <<P
import sys
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pipelined"]
import variantformer_amd.seq2gene.modules.layers as L
L.COUNTED_CONTEXT_KEYS = bool($flag)
import bench
bench.main()
P
python - <<P
import json
d=json.loads(open("gpurun_out/r5e/bench_c${flag}_$rep.json").read().strip().split("\n")[-1])
k=d["kernel_families"]
print("counted=$flag rep=$rep", d["value"], d["ms_per_step"], {n:k[n]["ms_per_step"] for n in ("attn/cre_ctx_cross","layernorm/cre_stream","gemm/cre_stream")})
P
done
done
