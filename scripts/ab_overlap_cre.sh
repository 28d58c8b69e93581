# A/B on one box: CRE layers on a side stream beside the gene layers (runtime.Switches.overlap_cre_stream, the default since round 6) vs serial
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for flag in 1 0; do
python - <<P 2>/dev/null | tail -1
import sys
sys.argv = ["bench.py", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-pipelined", "--no-kernel-timing"]
from variantformer_amd import runtime
runtime.set_for_this_context(overlap_cre_stream=bool($flag))
import bench, io, contextlib, json
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().split("\n")[-1])
print("overlap=$flag rep=$rep", d["value"], d["ms_per_step"], d.get("batch_of_8", {}).get("value"), d.get("batch_of_1", {}).get("value"))
P
done
done
python - <<P 2>&1 | tail -3
# same outputs either way (bit for bit), alerts merged
import torch, numpy as np
from tests.helpers import SEQ2REG_512, build_model, seq2gene_kw
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
import variantformer_amd.seq2gene.model_combined_modulator as M
model = build_model(SEQ2REG_512, seq2gene_kw(layers=5), seed=3).cuda()
batch = make_batch(4, [300, 40], [150, 20], [TISSUES_54[:5], [9, 33]], 200)
from variantformer_amd import runtime
with runtime.override(overlap_cre_stream=False):
    a = model.predict_step(batch, 0)
for _ in range(3): b = model.predict_step(batch, 0)
print("bit-identical:", all(np.array_equal(a[k][i], b[k][i]) for k in ("pred_gene_exp", "embeddings") for i in range(2)))
P
