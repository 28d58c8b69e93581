import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vf_oracle as O
from tests.conftest import load_fixture
from tests.helpers import build_model, state_dict_cpu
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
import bench

def stats(tag, a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    print(f"  {tag:34s} max-norm {d.max()/np.abs(b).max():.2e}  rms/rms {np.sqrt((d**2).mean())/np.sqrt((b**2).mean()):.2e}  frac>1e-3*max {(d > 1e-3*np.abs(b).max()).mean():.3f}")

def run(name, model, batch, sd, hps):
    print(name)
    outs = {}
    for t in ("1", "0"):
        os.environ["VF_TRUNK16"] = t
        outs[t] = model.predict_step(batch, 0)
    orc = {}
    for key, kw in (("t16", dict(trunk16=True)), ("f32trunk", dict(trunk16=False)), ("fp32", None)):
        rnd = None if kw is None else O.Rounding("bf16", **kw)
        orc[key] = O.predict_step(batch, sd, *hps, rounding=rnd if rnd is not None else None, share_cre_stream=True)
    for i in range(len(outs["1"]["pred_gene_exp"])):
        for t, k in (("1", "t16"), ("1", "f32trunk"), ("0", "f32trunk"), ("0", "t16"), ("1", "fp32"), ("0", "fp32")):
            stats(f"gene {i} emb  hip(t16={t}) vs orc({k})", outs[t]["embeddings"][i], orc[k]["embeddings"][i])
        for t, k in (("1", "t16"), ("0", "f32trunk"), ("1", "fp32"), ("0", "fp32")):
            stats(f"gene {i} expr hip(t16={t}) vs orc({k})", outs[t]["pred_gene_exp"][i], orc[k]["pred_gene_exp"][i])

meta, arrays, sd, batch = load_fixture("small_sin")
model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
run("small_sin", model, batch, sd, (hp, hp, O.Seq2GeneHP.from_kwargs(meta["seq2gene"])))
model, hp, kw = bench.build_model(torch.device("cuda", 0))
sd = state_dict_cpu(model)
batch = make_batch(4321, [24, 9], [6, 3], [TISSUES_54[:3], [62]], 200)
shp = O.Seq2RegHP.from_hparams(hp)
torch.set_num_threads(16)
run("full depth", model, batch, sd, (shp, shp, O.Seq2GeneHP.from_kwargs(kw)))
