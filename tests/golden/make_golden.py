#!/usr/bin/env python
"""Generate golden fixtures by running the *reference itself* on CPU in this container.

Runs only where /root/reference exists (the dev container); the fixtures it writes
(tests/golden/*.npz, *.json) are data: seeded inputs and the reference's outputs.
The Python reference never travels to the GPU box; these vectors do.

What is real and what is a stand-in:
  * real reference code executed: seq2reg.model.Seq2RegPredictor,
    seq2reg.modules.FlashTransformerLayer,
    seq2gene.model_combined_modulator.{Seq2GenePredictorCombinedModulator,CombinedModulator},
    seq2gene.modules.layers.{ContextFlashAttentionEncoderLayer,FlashAttLayer,MultiRegistry,
    TissueExpressionHeads}, utils.seq.BPEEncoder, utils.functions.precision2dtype.
  * stand-ins written here (third-party packages absent from the image):
    - lightning.pytorch.LightningModule -> nn.Module + save_hyperparameters()/log()
    - pybedtools -> empty module
    - flash_attn.modules.mha.MHA and flash_attn.bert_padding.{pad_input,unpad_input}
      (flash-attn v2.8.3, README.md:66 of the reference; CUDA only, not installable here):
      restated below from its published semantics: softmax(QK^T/sqrt(dh) - slope_h*|i-j|)V,
      non-causal, Wqkv packed "(three h d)", Wkv packed "(two h d)".
    => at the attention boundary parity is UNPINNED by anything runnable offline.

A fake trainer with precision "bf16-mixed" and no autocast selects the reference's
cast-free branch (model_combined_modulator.py:741-742), i.e. a clean fp32 run of the
reference's own orchestration.

Usage:  python tests/golden/make_golden.py      (writes next to this file)
Re-running reproduces the committed fixtures to the last fp32 ulp: 32 of the 36 files byte for byte, 4 differ by <= 1.5e-7 on
values near 0.7 (the CPU GEMMs' reduction order follows the machine's thread partition even with set_num_threads(4) pinned).
"""
from __future__ import annotations

import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# --------------------------------------------------------------------------------------
# stand-ins for absent third-party packages
# --------------------------------------------------------------------------------------
def _alibi_slopes(n):
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start**i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + _alibi_slopes(2 * c)[0::2][: n - c]


class _MHA(nn.Module):
    """Pure-PyTorch stand-in with flash_attn.modules.mha.MHA's parameter names."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, use_flash_attn=True, use_alibi=False,
                 cross_attn=False, **kw):
        super().__init__()
        self.embed_dim, self.num_heads, self.cross_attn = embed_dim, num_heads, cross_attn
        self.head_dim = embed_dim // num_heads
        if cross_attn:
            self.Wq = nn.Linear(embed_dim, embed_dim)
            self.Wkv = nn.Linear(embed_dim, 2 * embed_dim)
        else:
            self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.use_alibi = use_alibi
        if use_alibi:
            self.register_buffer("alibi_slopes", torch.tensor(_alibi_slopes(num_heads), dtype=torch.float32),
                                 persistent=False)

    def _attend(self, q, k, v):
        # q [sq,H,dh], k,v [sk,H,dh]; flash-attn's contract for 16-bit inputs: fp32 scores / softmax / accumulation, output in
        # the input dtype (a no-op for the fp32 fixtures; used by the precision="32" fixture, whose MHA modules run in fp16)
        odt = q.dtype
        q, k, v = q.float(), k.float(), v.float()
        return self._attend32(q, k, v).to(odt)

    def _attend32(self, q, k, v):
        s = torch.einsum("qhd,khd->hqk", q, k) / math.sqrt(self.head_dim)
        if self.use_alibi:
            sq, sk = q.shape[0], k.shape[0]
            i = torch.arange(sq)[:, None] + (sk - sq)
            j = torch.arange(sk)[None, :]
            s = s - self.alibi_slopes.to(s.dtype)[:, None, None] * (i - j).abs().to(s.dtype)[None]
        p = torch.softmax(s, dim=-1)
        return torch.einsum("hqk,khd->qhd", p, v)

    def forward(self, x, x_kv=None, cu_seqlens=None, max_seqlen=None, cu_seqlens_k=None, max_seqlen_k=None, **kw):
        H, dh = self.num_heads, self.head_dim
        if self.cross_attn:
            q = self.Wq(x)
            kv = self.Wkv(x_kv if x_kv is not None else x)
        else:
            qkv = self.Wqkv(x)
        if cu_seqlens is None:
            # padded [B,S,D], no masking
            B = x.shape[0]
            outs = []
            for b in range(B):
                if self.cross_attn:
                    qb = q[b].view(-1, H, dh)
                    kvb = kv[b].view(-1, 2, H, dh)
                    outs.append(self._attend(qb, kvb[:, 0], kvb[:, 1]).reshape(-1, H * dh))
                else:
                    t = qkv[b].view(-1, 3, H, dh)
                    outs.append(self._attend(t[:, 0], t[:, 1], t[:, 2]).reshape(-1, H * dh))
            return self.out_proj(torch.stack(outs))
        out = torch.empty(x.shape[0], H * dh, dtype=x.dtype)
        n = cu_seqlens.numel() - 1
        for b in range(n):
            a, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
            if self.cross_attn:
                ka, ke = int(cu_seqlens_k[b]), int(cu_seqlens_k[b + 1])
                kvb = kv[ka:ke].view(-1, 2, H, dh)
                out[a:e] = self._attend(q[a:e].view(-1, H, dh), kvb[:, 0], kvb[:, 1]).reshape(-1, H * dh)
            else:
                t = qkv[a:e].view(-1, 3, H, dh)
                out[a:e] = self._attend(t[:, 0], t[:, 1], t[:, 2]).reshape(-1, H * dh)
        return self.out_proj(out)


def _unpad_input(hidden_states, attention_mask, unused_mask=None):
    seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    max_len = int(seqlens.max().item())
    cu = torch.nn.functional.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(-1, *hidden_states.shape[2:])
    return flat[indices], indices, cu, max_len, seqlens


def _pad_input(hidden_states, indices, batch, seqlen):
    out = torch.zeros(batch * seqlen, *hidden_states.shape[1:], dtype=hidden_states.dtype)
    out[indices] = hidden_states
    return out.view(batch, seqlen, *hidden_states.shape[1:])


class _LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = None

    def save_hyperparameters(self, *a, ignore=None, **k):
        import inspect
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        hp = {n: args.locals[n] for n in args.args if n != "self"}
        if args.keywords:
            hp.update(args.locals[args.keywords])
        for n in (ignore or []):
            hp.pop(n, None)
        self.hparams = types.SimpleNamespace(**hp)

    def log(self, *a, **k):
        pass


def install_stubs():
    lightning = types.ModuleType("lightning")
    pl = types.ModuleType("lightning.pytorch")
    pl.LightningModule = _LightningModule
    lightning.pytorch = pl
    sys.modules["lightning"] = lightning
    sys.modules["lightning.pytorch"] = pl
    sys.modules["pybedtools"] = types.ModuleType("pybedtools")
    fa = types.ModuleType("flash_attn")
    fam = types.ModuleType("flash_attn.modules")
    mha = types.ModuleType("flash_attn.modules.mha")
    mha.MHA = _MHA
    bp = types.ModuleType("flash_attn.bert_padding")
    bp.pad_input, bp.unpad_input = _pad_input, _unpad_input
    fa.modules, fam.mha, fa.bert_padding = fam, mha, bp
    for n, m in [("flash_attn", fa), ("flash_attn.modules", fam), ("flash_attn.modules.mha", mha),
                 ("flash_attn.bert_padding", bp)]:
        sys.modules[n] = m
    # the reference's top-level packages (utils, datasets, ...) must win over site-packages
    for n in list(sys.modules):
        if n == "datasets" or n.startswith("datasets.") or n == "utils" or n.startswith("utils."):
            del sys.modules[n]
    sys.path.insert(0, REF)


# --------------------------------------------------------------------------------------
# fixture configs
# --------------------------------------------------------------------------------------
# name -> (seq2reg hparams, seq2gene kwargs, batch geometry)
FIXTURES = {
    # sinusoidal seq2reg (dh=64), modulator dh=48, ragged N / C / T, padded genes
    "small_sin": dict(
        seed=101,
        seq2reg=dict(vocab_size=500, embedding_dim=128, num_heads=2, num_layers=2, num_tissues=3,
                     num_classes=2, learning_rate=1e-4, loss_fn=["cross_entropy", "0"], seq_pool="mean",
                     cre_type="binary", token_length=40, use_context=False,
                     positional_encoding="sinusoidal", use_flash=True),
        seq2gene=dict(num_tissues=63, emb_dim=192, gene_emb_dim=128, num_heads=4, num_layers=3,
                      use_alibi=True, mlp_dout=0.1, use_context=True, token_dim=128,
                      gene_pooling="multi_registry", multi_head=False, use_bigger_head=True,
                      only_cross_attention=False, cross_alibi=False, add_context_to_cres=False,
                      use_res=False, train_gene_tokenizer=True, use_batching=True),
        n_cres=[5, 9], n_chunks=[3, 4], tissues=[[7, 20, 62], [33]], token_length=40,
        cre_len_range=(8, 30),
    ),
    # ALiBi seq2reg, 3 genes, more tissues, a gene with a single CRE
    "small_alibi": dict(
        seed=202,
        seq2reg=dict(vocab_size=500, embedding_dim=128, num_heads=2, num_layers=2, num_tissues=3,
                     num_classes=2, learning_rate=1e-4, loss_fn=["cross_entropy", "0"], seq_pool="mean",
                     cre_type="binary", token_length=32, use_context=False,
                     positional_encoding="alibi", use_flash=True),
        seq2gene=dict(num_tissues=63, emb_dim=96, gene_emb_dim=128, num_heads=2, num_layers=2,
                      use_alibi=True, mlp_dout=0.1, use_context=True, token_dim=128,
                      gene_pooling="multi_registry", multi_head=False, use_bigger_head=True,
                      only_cross_attention=False, cross_alibi=False, add_context_to_cres=False,
                      use_res=False, train_gene_tokenizer=True, use_batching=True),
        n_cres=[1, 12, 7], n_chunks=[2, 1, 5], tissues=[[10, 11], [7, 8, 9, 59], [62]], token_length=32,
        cre_len_range=(4, 33),
    ),
}


def build_reference_model(fx):
    from seq2reg.model import Seq2RegPredictor
    from seq2gene.model_combined_modulator import Seq2GenePredictorCombinedModulator
    if fx.get("model_class") == "Seq2GenePredictor":
        from seq2gene.model import Seq2GenePredictor as Seq2GenePredictorCombinedModulator  # noqa: F811
    from variantformer_amd.utils.synthetic import fill_state_dict

    cre_tok = Seq2RegPredictor(**fx["seq2reg"])
    gene_tok = Seq2RegPredictor(**fx["seq2reg"])
    model = Seq2GenePredictorCombinedModulator(cre_tokenizer=cre_tok, gene_tokenizer=gene_tok, **fx["seq2gene"])
    fill_state_dict(model, fx["seed"])
    model.eval()
    model.vep = False
    # "bf16-mixed": the cast-free fp32 branch (no autocast on CPU); "32": MHA modules in fp16, the rest fp32
    model.trainer = types.SimpleNamespace(precision=fx.get("precision", "bf16-mixed"))
    return model


def _with(base, **opts):
    fx = dict(FIXTURES[base])
    fx["seq2gene"] = dict(fx["seq2gene"], **opts)
    return fx


# options the shipped configuration leaves off: cross-attention-only gene layers, gene
# residual, ALiBi on the cross attentions, one shared start token / tissue embedding added to the CRE tokens, max pooling
FIXTURES["small_opts_a"] = dict(_with("small_sin", only_cross_attention=True, use_res=True, cross_alibi=True,
                                      gene_pooling="start_token"), seed=404)
FIXTURES["small_opts_b"] = dict(_with("small_alibi", add_context_to_cres=True, gene_pooling="max"), seed=505)
# per-tissue small MLP heads; context-free CRE layers with one shared linear head
FIXTURES["small_opts_c"] = dict(_with("small_sin", multi_head=True, use_bigger_head=False), seed=606)
FIXTURES["small_opts_d"] = dict(_with("small_alibi", use_context=False, head_type="linear", use_bigger_head=False), seed=707)
# trainer.precision = "32" (reference utils/functions.py:28-30 -> torch.float32 -> layers.py:98-126: the MHA modules are cast
# to fp16 for their forward -- weights included, in place -- and everything else stays fp32)
FIXTURES["small_sin_p32"] = dict(FIXTURES["small_sin"], seed=111, precision="32")
FIXTURES["small_twomod"] = dict(FIXTURES["small_sin"], seed=303, model_class="Seq2GenePredictor",
                                n_cres=[6, 3], n_chunks=[2, 4], tissues=[[62, 7], [20, 33, 59]])
# the older class with ITS constructor defaults' gene layers (only_cross_attention=True, reference layers.py:753) and ALiBi on the
# cross attention (use_res=True fails inside the reference itself there: padded `res` + unpadded stream, layers.py:912);
# and with context-free CRE layers (use_context=False, layers.py:612-624)
FIXTURES["small_twomod_b"] = dict(_with("small_alibi", only_cross_attention=True, cross_alibi=True), seed=909,
                                  model_class="Seq2GenePredictor", n_cres=[4, 9, 2], n_chunks=[3, 1, 2],
                                  tissues=[[5, 6], [62], [7, 30, 31]])
FIXTURES["small_twomod_c"] = dict(_with("small_sin", use_context=False), seed=1010, model_class="Seq2GenePredictor",
                                  n_cres=[7, 2], n_chunks=[2, 3], tissues=[[1, 2, 3], [40]])


def run_fixture(name, fx):
    from variantformer_amd.utils.synthetic import make_batch

    torch.manual_seed(0)
    torch.set_num_threads(4)
    model = build_reference_model(fx)
    batch = make_batch(fx["seed"], fx["n_cres"], fx["n_chunks"], fx["tissues"], fx["token_length"],
                       cre_len_range=fx["cre_len_range"])
    if not hasattr(model, "combined_modulator"):
        batch["cre_attention_mask"] = batch["cre_attention_masks"]   # key the older class reads (seq2gene/model.py:656)
    captured = {}

    # capture seq2reg embeddings and modulator output through forward hooks (no reference edits)
    def hook_tok(tag):
        def _h(mod, inp, out):
            captured.setdefault(tag, []).append(out.detach().clone())
        return _h
    h1 = model.cre_tokenizer.register_forward_hook(hook_tok("cre_tok"))
    h2 = model.gene_tokenizer.register_forward_hook(hook_tok("gene_tok"))
    combined = hasattr(model, "combined_modulator")
    layer_out = []
    if combined:
        h3 = model.combined_modulator.register_forward_hook(
            lambda m, i, o: captured.__setitem__("modulator_gene_out", o[0].detach().clone()))
        hs = [l.register_forward_hook(lambda m, i, o: layer_out.append(o.detach().clone()))
              for l in list(model.combined_modulator.gene_layers) + list(model.combined_modulator.cre_layers)]
    else:
        h3 = model.gene_modulator.register_forward_hook(lambda m, i, o: None)
        hs = []
    with torch.no_grad():
        out = model.predict_step(batch, 0)
    for h in [h1, h2, h3] + hs:
        h.remove()

    arrays = {}
    for i, (p, e) in enumerate(zip(out["pred_gene_exp"], out["embeddings"])):
        arrays[f"pred_gene_exp_{i}"] = np.asarray(p, np.float32)
        arrays[f"embeddings_{i}"] = np.asarray(e, np.float32)
    for i, t in enumerate(captured["cre_tok"]):
        arrays[f"cre_tok_{i}"] = t.numpy()
    for i, t in enumerate(captured["gene_tok"]):
        arrays[f"gene_tok_{i}"] = t.numpy()
    if combined:
        arrays["modulator_gene_out"] = captured["modulator_gene_out"].numpy()
        # order of layer hooks firing: gene0, (cre_i, gene_{i+1})*
        arrays["first_gene_layer_out"] = layer_out[0].numpy()
        arrays["first_cre_layer_out"] = layer_out[1].numpy()
    # state-dict inventory (names + shapes) and a checksum of the generated weights
    sd = model.state_dict()
    inv = {k: list(v.shape) for k, v in sd.items()}
    chk = float(sum(float(v.double().abs().sum()) for v in sd.values() if torch.is_floating_point(v)))
    meta = dict(name=name, seed=fx["seed"], seq2reg=fx["seq2reg"], seq2gene=fx["seq2gene"], precision=fx.get("precision", "bf16-mixed"),
                model_class=fx.get("model_class", "Seq2GenePredictorCombinedModulator"),
                n_cres=fx["n_cres"], n_chunks=fx["n_chunks"], tissues=fx["tissues"],
                token_length=fx["token_length"], cre_len_range=list(fx["cre_len_range"]),
                state_dict_shapes=inv, weight_abs_sum=chk,
                generated_by="tests/golden/make_golden.py against /root/reference (fp32, CPU, stubbed flash_attn)")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(f"[golden] {name}: pred", [a.ravel()[:3] for k, a in arrays.items() if k.startswith("pred")])


VEP_MODEL = dict(base="small_sin", seed=808, n_cre=7, n_chunks=5, tissues=[7, 20, 62], cre_index=2, gene_index=[1, 1, 3])
# the same call on option sets the shipped configuration leaves off: (a) cross-attention-only gene layers, gene residual,
# ALiBi on the cross attention, one shared start token; (b) tissue embedding added to the CRE tokens, max pooling (no token in
# front of the gene chunks: the +1 of :665-666 does not apply)
VEP_MODEL_OPTS = {
    "vep_model_opts_a": dict(base="small_opts_a", seed=818, n_cre=6, n_chunks=4, tissues=[20, 62], cre_index=1, gene_index=[0, 2, 2]),
    "vep_model_opts_b": dict(base="small_opts_b", seed=828, n_cre=5, n_chunks=3, tissues=[7, 8, 9], cre_index=4, gene_index=[1, 1, 1]),
}


def vep_model_opts_fixture():
    for name, v in VEP_MODEL_OPTS.items():
        vep_model_fixture(name, v)


def vep_model_fixture(name="vep_model", v=None):
    """The reference's own variant_prediction (seq2gene/model_combined_modulator.py:909-1004) on a ref / het / hom
    batch with token positions: pred_gene_exp, embd and the token-position gathers (:296-326, +1 for the registry
    token :665-666).  Also one run with NaN positions (the reference then returns zero token embeddings, :936-939)."""
    from variantformer_amd.utils.synthetic import make_vep_batch
    v = VEP_MODEL if v is None else v
    fx = dict(FIXTURES[v["base"]], seed=v["seed"])
    torch.manual_seed(0)
    model = build_reference_model(fx)
    model.vep = True
    vb = make_vep_batch(v["seed"], v["n_cre"], v["n_chunks"], v["tissues"], fx["token_length"], cre_index=v["cre_index"],
                        gene_index=tuple(v["gene_index"]), cre_len_range=fx["cre_len_range"])
    arrays = {}
    with torch.no_grad():
        out = model.predict_step(vb, 0)
        nan_b = dict(vb, cre_token_position=torch.full((3, 1), float("nan")), gene_token_position=torch.full((3, 1), float("nan")))
        out_nan = model.predict_step(nan_b, 0)
    assert set(out) == {"pred_gene_exp", "embd", "variant_type", "gene_token_embedding", "cre_token_embedding"}
    for tag, o in (("pos", out), ("nan", out_nan)):
        for k in ("pred_gene_exp", "embd", "gene_token_embedding", "cre_token_embedding"):
            for i, a in enumerate(o[k]):
                arrays[f"{tag}.{k}_{i}"] = np.asarray(a, np.float32)
    sd = model.state_dict()
    meta = dict(v, seq2reg=fx["seq2reg"], seq2gene=fx["seq2gene"], token_length=fx["token_length"],
                cre_len_range=list(fx["cre_len_range"]), state_dict_shapes={k: list(t.shape) for k, t in sd.items()},
                weight_abs_sum=float(sum(float(t.double().abs().sum()) for t in sd.values() if torch.is_floating_point(t))),
                variant_type=out["variant_type"],
                generated_by="tests/golden/make_golden.py: reference Seq2GenePredictorCombinedModulator.variant_prediction "
                             "(fp32, CPU, stubbed flash_attn) on variantformer_amd.utils.synthetic.make_vep_batch")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(f"[golden] {name}: gene_tok", arrays["pos.gene_token_embedding_1"][0, :3], "cre_tok", arrays["pos.cre_token_embedding_2"][0, :3],
          "nan ->", float(np.abs(arrays["nan.gene_token_embedding_0"]).max()))


S2R_OPTS = {
    # name: (overrides of the small_sin seq2reg hparams, seed)
    "ctx_max": (dict(use_context=True, expand_context=False, seq_pool="max"), 911),
    "ctx_expand_linear_alibi": (dict(use_context=True, expand_context=True, seq_pool="linear", positional_encoding="alibi"), 912),
    "linear": (dict(seq_pool="linear"), 913),
    "dh96": (dict(embedding_dim=192, num_heads=2), 914),
    "dh128_alibi": (dict(embedding_dim=256, num_heads=2, positional_encoding="alibi", seq_pool="max"), 915),
}


def s2r_opts_fixture():
    """The reference's own Seq2RegPredictor.forward(only_embed=True) (seq2reg/model.py:193-279) for the tokenizer
    options the shipped checkpoint may or may not use: context embedding with / without expand_context
    (seq2reg/modules.py:40-126), max / linear sequence pooling, head dims 96 / 128."""
    from seq2reg.model import Seq2RegPredictor
    from variantformer_amd.utils.synthetic import fill_state_dict, make_gene
    base = FIXTURES["small_sin"]["seq2reg"]
    arrays, meta = {}, {}
    for name, (over, seed) in S2R_OPTS.items():
        hp = dict(base, **over)
        torch.manual_seed(0)
        m = Seq2RegPredictor(**hp)
        fill_state_dict(m, seed)
        m.eval()
        g = make_gene(seed, 9, 2, [7], hp["token_length"], cre_len_range=(3, 41))
        with torch.no_grad():
            out = m(g["cre_sequences"], g["cre_attention_masks"], None, context=g["ref_cre_labels"], only_embed=True,
                    precision=None)
        arrays[name] = out.numpy()
        sd = m.state_dict()
        meta[name] = dict(hparams=hp, seed=seed, n_windows=9, cre_len_range=[3, 41],
                          state_dict_shapes={k: list(v.shape) for k, v in sd.items()},
                          weight_abs_sum=float(sum(float(v.double().abs().sum()) for v in sd.values() if torch.is_floating_point(v))))
        print(f"[golden] s2r_opts {name}: {tuple(out.shape)} {out.ravel()[:3].numpy()}")
    np.savez_compressed(os.path.join(HERE, "s2r_opts.npz"), **arrays)
    with open(os.path.join(HERE, "s2r_opts.json"), "w") as f:
        json.dump(dict(cases=meta, generated_by="tests/golden/make_golden.py: reference seq2reg.model.Seq2RegPredictor "
                                                "(fp32, CPU, stubbed flash_attn)"), f, indent=1, sort_keys=True)


def bpe_fixture():
    """Token-id golden vectors from the reference's BPEEncoder (utils/seq.py:8-62)."""
    from utils.seq import BPEEncoder
    from variantformer_amd.utils.synthetic import randint

    enc = BPEEncoder()
    enc.load_vocabulary(os.path.join(REF, "vocabs", "bpe_vocabulary_500.json"))
    alphabet = "ACGT"
    het = "RYSWKM"
    seqs = ["ACGTNNNNACGTRYACGT", "A", "acgtacgtnnacg", "NNNN", "ACGTBDHVACGT", "TTTTTTTTTTTTTTTTTTTT",
            "GATTACAGATTACAGATTACA", "N", "ANA", "CGCGCGCGCGCGCGCGCGCGCGCGCGCGCGCG"]
    for k in range(24):
        n = [17, 64, 150, 251, 350, 450, 1000, 5000][k % 8]
        r = randint(n, 0, 1000, 777 + k, 1)
        s = []
        for v in r:
            if v < 3:
                s.append("N")
            elif v < 8:
                s.append(het[int(v) % 6])
            else:
                s.append(alphabet[int(v) % 4])
        seqs.append("".join(s))
    out = []
    for s in seqs:
        ids, toks, ids_r, _ = enc.encode([s, "A"])
        out.append({"seq": s, "ids": [int(i) for i in ids]})
    # encode_with_position known answers (utils/seq.py:68-174)
    pos_cases = []
    for s in seqs[10:20]:
        for p in (0, len(s) // 3, len(s) - 1):
            try:
                r = enc.encode_with_position(s, p)
                pos_cases.append({"seq_index": seqs.index(s), "position": p,
                                  "result": json.loads(json.dumps(r, default=lambda o: o if isinstance(o, (int, float, str)) else list(o)))})
            except Exception as ex:  # record the reference's error behaviour too
                pos_cases.append({"seq_index": seqs.index(s), "position": p, "error": type(ex).__name__})
    with open(os.path.join(HERE, "bpe_ids.json"), "w") as f:
        json.dump({"cases": out, "position_cases": pos_cases,
                   "generated_by": "reference utils/seq.BPEEncoder + vocabs/bpe_vocabulary_500.json (tokenizers %s)"
                   % __import__("tokenizers").__version__}, f)
    print(f"[golden] bpe: {len(out)} strings; known answer:", out[0]["ids"])


def misc_fixture():
    """ALiBi slopes, sinusoidal PE, precision2dtype and pad/unpad conventions from the reference."""
    from seq2gene.modules.layers import get_alibi_slopes
    from seq2reg.model import positionalencoding1d
    from utils.functions import precision2dtype

    arrays = {}
    for h in (2, 4, 8, 12, 32):
        arrays[f"alibi_{h}"] = get_alibi_slopes(h).numpy().astype(np.float64)
    arrays["pe_128_40"] = positionalencoding1d(128, 40).numpy()
    arrays["pe_512_200"] = positionalencoding1d(512, 200).numpy()
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **arrays)
    prec = {}
    for s in ["bf16-mixed", "16-mixed", "32", "32-true", "bf16", "16", "64"]:
        try:
            prec[s] = str(precision2dtype(s))
        except Exception as ex:
            prec[s] = "ERR:" + type(ex).__name__
    with open(os.path.join(HERE, "misc.json"), "w") as f:
        json.dump({"precision2dtype": prec}, f, indent=1, sort_keys=True)
    print("[golden] misc:", prec)


def vep_fixture():
    """ref/het/hom batches from the reference's VEPDataset (datasets/vepdataset.py:134-799) on synthetic genome
    artifacts (tests/vep_artifacts.py).  duckdb (imported by utils/assets.py for the S3 manifests, unused here)
    is absent from the image and stubbed with an empty module; the manifests are plain path tables."""
    sys.modules.setdefault("duckdb", types.ModuleType("duckdb"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import vep_artifacts as va
    from datasets.vepdataset import VEPDataset, Variant
    from utils.seq import BPEEncoder
    import tempfile

    enc = BPEEncoder()
    enc.load_vocabulary(os.path.join(REF, "vocabs", "bpe_vocabulary_500.json"))
    spec = va.make_spec()

    def encode_pair(fwd, rev):
        f, _, r, _ = enc.encode([fwd, rev])
        return f, r

    class Table:
        def __init__(self, t):
            self.t = t

        def get_file_path(self, *key):
            return self.t[key[0] if len(key) == 1 else tuple(key)]

    arrays, cases = {}, []
    with tempfile.TemporaryDirectory() as root:
        gene_csv, gene_npz, cre_pkl = va.write_artifacts(spec, root, encode_pair)
        genes = {g["gene_id"]: g for g in spec["genes"]}
        pairs = [{"variant": Variant(chrom=v["chrom"], pos=v["pos"], ref=v["ref"], alt=v["alt"], tissue=v["tissue"],
                                     gene_id=[v["gene"]]),
                  "gene": genes[v["gene"]], "sample_name": v["sample_name"], "population": v["population"],
                  "vcf_path": None} for v in spec["variants"]]
        ds = VEPDataset(enc, Table(gene_csv), Table(gene_npz), Table(cre_pkl), gene_variant_pairs=pairs,
                        **spec["settings"])
        for i in range(len(ds)):
            try:
                batch = ds[i]
            except Exception as ex:
                cases.append({"error": type(ex).__name__})
                continue
            cases.append({"variant_type": batch["variant_type"]})
            for k, a in va.flatten_batch(batch).items():
                arrays[f"case{i}.{k}"] = a
        # string-level known answers of SequenceProcessor (:37-131)
        from datasets.vepdataset import SequenceProcessor as SP
        seqs = ["ACGTNRYSWKMBDHVacgtn-.", "GATTACA", "x?ACGU"]
        sp = {"reverse_complement": {s: SP.reverse_complement(s) for s in seqs},
              "iupac": {a + b: SP.get_iupac_code(a, b) for a in "ACGTNa" for b in "ACGTNa"},
              "apply": [list(SP.apply_variant("ACGTAC,GTACGT", Variant("1", 1, r, a, [0], ["g"]), p))
                        for r, a, p in (("G", "T", 2), ("A", "A", 0), ("C", "N", 5))]}
    spec["cases"], spec["sequence_processor"] = cases, sp
    spec["generated_by"] = "tests/golden/make_golden.py: reference datasets/vepdataset.VEPDataset on tests/vep_artifacts.py files"
    np.savez_compressed(os.path.join(HERE, "vep.npz"), **arrays)
    with open(os.path.join(HERE, "vep.json"), "w") as f:
        json.dump(spec, f, indent=1)
    print("[golden] vep:", [c.get("variant_type", c.get("error")) for c in cases])


def variantprocessor_fixture():
    """Output stage of the reference's VariantProcessor (processors/variantprocessor.py:303-497) and its log2fc scores
    (utils/functions.py:184-301) on seeded predictions.  omegaconf (absent) is only used by __init__, which is not
    run: the three methods are called unbound on a plain namespace carrying the attributes they read."""
    import tempfile
    sys.modules.setdefault("duckdb", types.ModuleType("duckdb"))
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", oc)
    sys.modules["lightning.pytorch"].Trainer = object
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import vep_artifacts as va
    from datasets.vepdataset import Variant
    from processors.variantprocessor import VariantProcessor as RefVP
    from utils.functions import generate_log2fc_score
    import yaml
    with open(os.path.join(REF, "vocabs", "tissue_vocab.yaml")) as f:
        tissue_vocab = yaml.safe_load(f)
    arrays = {}
    with tempfile.TemporaryDirectory() as root:
        for tag, with_sample in (("pop", False), ("sample", True)):
            pairs, preds, af = va.make_vp_case(Variant, with_sample=with_sample)
            af_dir = va.write_af_tables(af, os.path.join(root, "af"))
            me = types.SimpleNamespace(gene_variant_pairs=pairs, config=types.SimpleNamespace(emb_dim=6, af_path=af_dir),
                                       tissue_idx_to_name={v: k for k, v in tissue_vocab.items()},
                                       _get_variant_output_path=lambda: os.path.join(root, f"{tag}_VF.parquet"))
            long_df = RefVP.compile_predictions(me, preds, vcf_path="x.vcf.gz" if with_sample else None)
            wide_df = RefVP.format_scores(me, long_df.copy())
            score_df = generate_log2fc_score(wide_df.copy(), af_dir)
            for name, df in (("long", long_df), ("wide", wide_df), ("score", score_df)):
                arrays.update(va.frame_to_arrays(df, f"{tag}.{name}"))
            print(f"[golden] variantprocessor {tag}: long {long_df.shape} wide {wide_df.shape} score {score_df.shape}")
    np.savez_compressed(os.path.join(HERE, "variantprocessor.npz"), **arrays)


def layer_data_kv_fixture():
    """make_data_kv=True: the option swaps the two streams of a layer's cross attention (queries from the raw context, keys /
    values from the normalised stream; seq2gene/modules/layers.py:133-136,283-286, seq2reg/modules.py:97-100).  No reference
    MODEL can enable it (the models build their layers with the default), so it is pinned at layer level: the reference's own
    three layer classes, seeded weights, one padded batch with ragged masks, fp32 (precision=None: no casts)."""
    from seq2gene.modules.layers import ContextFlashAttentionEncoderLayer as GeneLayer
    from seq2gene.modules.layers import ContextFlashCrossAttentionEncoderLayer as CrossLayer
    from seq2reg.modules import ContextFlashAttentionEncoderLayer as S2RLayer
    g = torch.Generator().manual_seed(20251207)
    D, H, F, B, S = 128, 4, 256, 3, 24
    src = torch.randn((B, S, D), generator=g)
    ctx = torch.randn((B, S, D), generator=g)
    mask = torch.zeros((B, S), dtype=torch.bool)                 # True = pad
    mask[0, 20:] = True
    mask[1, 7:] = True
    arrays = {"src": src.numpy(), "ctx": ctx.numpy(), "mask": mask.numpy()}
    meta = {"d_model": D, "nhead": H, "hidden_dim": F, "layers": {}}
    layers = {
        "gene": (GeneLayer(D, H, hidden_dim=F, dropout=0.0, use_alibi=True, make_data_kv=True, mlp_dout=0.0),
                 lambda m: m(src.clone(), ctx.clone(), src_key_padding_mask=mask.clone(), precision=None), True),
        "cross": (CrossLayer(D, H, hidden_dim=F, dropout=0.0, use_alibi=False, make_data_kv=True, mlp_dout=0.0),
                  lambda m: m(src.clone(), ctx.clone(), context_padding_mask=mask.clone(), src_key_padding_mask=mask.clone(),
                              precision=None), False),
        "s2r": (S2RLayer(D, H, hidden_dim=F, dropout=0.0, use_alibi=False, make_data_kv=True, mlp_dout=0.0),
                lambda m: m(src.clone(), ctx.clone(), key_padding_mask=mask.clone(), precision=None), False),
    }
    for name, (layer, call, alibi) in layers.items():
        layer.eval()
        with torch.no_grad():
            for k, p in layer.named_parameters():               # non-trivial LayerNorm affine parameters, O(1) activations
                if "norm" in k:
                    p.copy_((1.0 if k.endswith("weight") else 0.0) + 0.2 * torch.randn(p.shape, generator=g))
                elif k.endswith("bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p.shape[-1]))
            out = call(layer)
        for k, v in layer.state_dict().items():
            arrays[f"{name}.sd.{k}"] = v.numpy()
        arrays[f"{name}.out"] = out.numpy()
        meta["layers"][name] = {"use_alibi": alibi, "class": type(layer).__module__ + "." + type(layer).__name__}
        print(f"[golden] layer_data_kv {name}: out {tuple(out.shape)} |out| {float(out.abs().mean()):.3f}")
    np.savez_compressed(os.path.join(HERE, "layer_data_kv.npz"), **arrays)
    with open(os.path.join(HERE, "layer_data_kv.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    assert os.path.isdir(REF), "reference checkout not present: fixtures can only be generated in the dev container"
    sys.path.insert(0, REPO)
    import variantformer_amd.utils.synthetic  # noqa: F401  (import ours before the reference's `utils`)
    install_stubs()
    import seq2reg.model  # noqa: F401  (its import sets torch.set_float32_matmul_precision("medium"), seq2reg/model.py:12)
    # "medium" lets oneDNN run fp32 matmuls through bf16 on this CPU (observed: 2e-3 error on the
    # seq2reg embeddings), which would make the fixtures a noisy fp32 reference.  The flag is a
    # speed/precision knob, not part of the algorithm: restore exact fp32 for the golden run.
    torch.set_float32_matmul_precision("highest")
    only = sys.argv[1:]
    if only:                      # e.g. `make_golden.py vep_model`: regenerate selected fixtures only
        for name in only:
            if name in FIXTURES:
                run_fixture(name, FIXTURES[name])
            else:
                globals()[f"{name}_fixture"]()
        return
    for name, fx in FIXTURES.items():
        run_fixture(name, fx)
    vep_model_fixture()
    vep_model_opts_fixture()
    s2r_opts_fixture()
    bpe_fixture()
    misc_fixture()
    vep_fixture()
    variantprocessor_fixture()
    layer_data_kv_fixture()


if __name__ == "__main__":
    main()
