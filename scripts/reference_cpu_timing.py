"""Time the REFERENCE's own PyTorch path on this container's host cores at BASELINE configs[0] geometry
(single gene, 128 kb window, 1 tissue: N = 40 cCRE windows, C = 178 gene chunks, T = 1; SURVEY 8d cfg 1), full
1.2B-parameter architecture with random weights.  Dev container only (/root/reference must exist); the third-party
packages the reference needs and the image lacks are the in-memory stand-ins of tests/golden/make_golden.py
(lightning, pybedtools, flash_attn -> plain PyTorch attention).  Result recorded in BASELINE.md.

    python scripts/reference_cpu_timing.py
"""
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import torch  # noqa: E402

import variantformer_amd.utils.synthetic as synthetic  # noqa: E402  (ours, before the reference's `utils`)
import bench  # noqa: E402
import make_golden as mg  # noqa: E402

assert os.path.isdir(mg.REF), "reference checkout not present"
mg.install_stubs()
from seq2reg.model import Seq2RegPredictor  # noqa: E402
from seq2gene.model_combined_modulator import Seq2GenePredictorCombinedModulator  # noqa: E402

torch.set_float32_matmul_precision("highest")
threads = len(os.sched_getaffinity(0))
torch.set_num_threads(threads)
torch.manual_seed(0)
cre_tok, gene_tok = Seq2RegPredictor(**bench.SEQ2REG_HP), Seq2RegPredictor(**bench.SEQ2REG_HP)
model = Seq2GenePredictorCombinedModulator(cre_tokenizer=cre_tok, gene_tokenizer=gene_tok, **bench.SEQ2GENE_KW)
model.eval()
model.vep = False
model.trainer = types.SimpleNamespace(precision="bf16-mixed")      # cast-free fp32 branch (no autocast on CPU)
n_params = sum(p.numel() for p in model.parameters())
batch = synthetic.make_batch(1281, [40], [178], [[33]], 200)
with torch.no_grad():
    t0 = time.perf_counter()
    out = model.predict_step(batch, 0)
    t1 = time.perf_counter()
    out = model.predict_step(batch, 0)
    t2 = time.perf_counter()
print(f"reference CPU path, cfg 1 (N=40, C=178, T=1), {n_params / 1e9:.2f} B parameters, fp32, {threads} threads: "
      f"first call {t1 - t0:.2f} s, second call {t2 - t1:.2f} s -> {1.0 / (t2 - t1):.3f} genes/s; "
      f"pred {out['pred_gene_exp'][0].ravel()}")
