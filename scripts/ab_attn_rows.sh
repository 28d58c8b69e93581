set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "row_map or kernel_independent or one_block" > gpurun_out/r5b/t1.log 2>&1; tail -3 gpurun_out/r5b/t1.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lookup or row_map or dedup" > gpurun_out/r5b/t2.log 2>&1; tail -3 gpurun_out/r5b/t2.log
for rep in 1 2; do
for flag in 1 0; do
python - > gpurun_out/r5b/bench_rows${flag}_$rep.json 2> gpurun_out/r5b/bench_rows${flag}_$rep.err <<P
import sys
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
from variantformer_amd import runtime
runtime.set_for_this_context(rows_in_attention=bool($flag))
import bench
bench.main()
P
python - <<P
import json
d=json.loads(open("gpurun_out/r5b/bench_rows${flag}_$rep.json").read().strip().split("\n")[-1])
k=d["kernel_families"]
print("rows=$flag rep=$rep", d["value"], d["ms_per_step"], {n:k[n]["ms_per_step"] for n in ("attn/seq2reg_self","attn/gene_self","layernorm/seq2reg","layernorm/gene_stream")})
P
done
done
