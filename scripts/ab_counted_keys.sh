# A/B on one box of the CRE layers' context cross attention: low-rank form (two skinny GEMMs around a 9-way softmax; default),
# counted-key attention kernel, and the round-4 form over the gathered [N, 2D] rows
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "counted or row_map or lowrank" > gpurun_out/r5e/t1.log 2>&1; tail -3 gpurun_out/r5e/t1.log
python -m pytest tests/test_model_gpu.py tests/test_ln_heal_gpu.py tests/test_trained_like_gpu.py -x -q -m gpu > gpurun_out/r5e/t2.log 2>&1; tail -3 gpurun_out/r5e/t2.log
for rep in 1 2; do
for form in lowrank counted expanded; do
python - > gpurun_out/r5e/bench_${form}_$rep.json 2> gpurun_out/r5e/bench_${form}_$rep.err <<P
import sys
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pipelined"]
from variantformer_amd import runtime
runtime.set_for_this_context(counted_context_keys="$form" != "expanded", lowrank_context="$form" == "lowrank")
import bench
bench.main()
P
python - <<P
import json
d=json.loads(open("gpurun_out/r5e/bench_${form}_$rep.json").read().strip().split("\n")[-1])
k=d["kernel_families"]
print("form=$form rep=$rep", d["value"], d["ms_per_step"], {n:k[n]["ms_per_step"] for n in ("attn/cre_ctx_cross","layernorm/cre_stream","gemm/cre_stream")})
P
done
done
