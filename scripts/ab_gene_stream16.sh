# A/B on one box: the first gene layer's input built from 16-bit copies of the distinct rows (default) against the fp32 row gather
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d
python -m pytest tests/test_model_gpu.py tests/test_ln_heal_gpu.py -x -q -m gpu > gpurun_out/r5d/t1.log 2>&1; tail -3 gpurun_out/r5d/t1.log
python -m pytest tests/test_trained_like_gpu.py -x -q -s -m gpu 2>&1 | grep -E "trained-like|passed|failed|signal"
for rep in 1 2; do
for flag in 1 0; do
python - > gpurun_out/r5d/bench_s${flag}_$rep.json 2> gpurun_out/r5d/bench_s${flag}_$rep.err <<P
import sys
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pipelined"]
import variantformer_amd.seq2gene.modules.layers as L
if not $flag:
    orig = L.ContextFlashAttentionEncoderLayer.self_qkv_of_unique_rows
    def no_stream(self, *a, with_stream=False, **k):
        r = orig(self, *a, with_stream=with_stream, **k)
        return (r[0], None) if (with_stream and r is not None) else r
    L.ContextFlashAttentionEncoderLayer.self_qkv_of_unique_rows = no_stream
import bench
bench.main()
P
python - <<P
import json
d=json.loads(open("gpurun_out/r5d/bench_s${flag}_$rep.json").read().strip().split("\n")[-1])
k=d["kernel_families"]
print("stream16=$flag rep=$rep", d["value"], d["ms_per_step"], {n:k[n]["ms_per_step"] for n in ("layernorm/gene_stream","gemm/gene_stream")})
P
done
done
