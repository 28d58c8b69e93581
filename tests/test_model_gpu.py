"""End-to-end parity on a real MI355X, through the reference's API surface (predict_step on a
collate_fn_batching dict):

  1. against the golden fixtures = outputs of the reference's own classes (fp32):  bf16-operand kernels vs an
     fp32 run differ at bf16 level, tolerance BF16_VS_FP32 below;
  2. against the CPU oracle evaluated with the kernels' rounding points (oracle rounding="bf16"): the two then
     differ only by accumulation order / exp ulps -> the north-star tolerance of 1e-3 relative.
"""
import numpy as np
import pytest
import torch

from oracle import vf_oracle as O
from tests.conftest import load_fixture
from tests.helpers import SEQ2REG_512, build_model, check_signal, prel, seq2gene_kw, state_dict_cpu
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

pytestmark = pytest.mark.gpu

NORTH_STAR_RTOL = 1e-3      # BASELINE.json: "within 1e-3 relative fp32 tolerance" (vs same-rounding oracle)
BF16_VS_FP32 = 2e-2         # bf16-operand arithmetic vs the reference's fp32 fixture values
# Embeddings of the headline-size gene against PURE fp32 arithmetic (test_headline_size_gene_vs_oracle_and_properties): limits =
# 1.5 x the values MI355X measured (profiles/r06_c_headline_accuracy.log); max-norm = max |err| / max |ref|, element-wise =
# max |err| / (|ref| + rms(ref)).
HEADLINE_EMB_MAXNORM = {"bf16": 6e-3, "fp16": 3e-3}          # measured 3.93e-3 / 1.97e-3
HEADLINE_EMB_ELEMENTWISE = {"bf16": 2.2e-2, "fp16": 6e-3}    # measured 1.48e-2 / 3.71e-3


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _erel(a, b):
    """element-wise: max |a-b| / (|b| + rms(b))"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / (np.abs(b) + np.sqrt((b * b).mean()) + 1e-30)).max())


def _hps(meta):
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    return hp, hp, O.Seq2GeneHP.from_kwargs(meta["seq2gene"])


def test_predict_step_vs_reference_golden(golden):
    meta, arrays, sd, batch = golden
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    out = model.predict_step(batch, 0)
    cre_hp, gene_hp, hp = _hps(meta)
    orc = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding="bf16", share_cre_stream=True)
    assert out["batch_idx"] == 0 and out["dataloader_idx"] is None
    for i in range(len(meta["n_cres"])):
        p, e = out["pred_gene_exp"][i], out["embeddings"][i]
        assert p.dtype == np.float32 and p.shape == (len(meta["tissues"][i]), 1)
        assert e.shape == (len(meta["tissues"][i]), meta["seq2gene"]["emb_dim"])
        # expression output vs the REFERENCE's fp32 value: inside the north-star tolerance directly
        assert prel(p, arrays[f"pred_gene_exp_{i}"]) < NORTH_STAR_RTOL
        assert _rel(e, arrays[f"embeddings_{i}"]) < 5e-3               # bf16-operand noise on a D-wide vector (max norm)
        assert prel(p, orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(e, orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL     # 1536-wide vector: max over many elements
    n = len(meta["n_cres"])
    check_signal("golden fixture vs reference fp32", out["pred_gene_exp"], [arrays[f"pred_gene_exp_{i}"] for i in range(n)])
    check_signal("golden fixture vs same-rounding oracle", out["pred_gene_exp"], orc["pred_gene_exp"])


def test_seq2reg_embeddings_vs_reference_golden(golden):
    meta, arrays, sd, batch = golden
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    for i in range(len(meta["n_cres"])):
        got = model.cre_tokenizer(batch["cre_sequences"][i], batch["cre_attention_masks"][i], None, only_embed=True)
        assert got.shape == arrays[f"cre_tok_{i}"].shape
        assert _rel(got.cpu().numpy(), arrays[f"cre_tok_{i}"]) < BF16_VS_FP32
        got = model.gene_tokenizer(batch["gene_embeddings"][i], batch["gene_attention_masks"][i], None, only_embed=True)
        assert _rel(got.cpu().numpy(), arrays[f"gene_tok_{i}"]) < BF16_VS_FP32


@pytest.mark.parametrize("precision", ["bf16-mixed", "16-mixed"])
@pytest.mark.parametrize("ln_fold", [True, False])
def test_seq2reg_mean_pool_before_the_last_down_projection(precision, ln_fold):
    """The encoder's mean pool is taken BEFORE the last layer's down-projection (mean(src + W2 h + b) = mean(src) + W2 mean(h) +
    b: FlashTransformerLayer._pooled_down_projection, vf_segment_mean16) -- an exact re-ordering.  Production width (d = 512, 8
    heads, GeGLU 2048 -> 1024), ragged windows of 1 ... 200 tokens and an empty one: the pooled embeddings agree with the
    reference's order of operations (token rows of the last layer, then the pool) to fp32 summation-order level, 2e-5 of the
    embedding scale, with the LayerNorm fold on and off (the self-healing path), bf16 and fp16 operands; the empty window is
    NaN both ways (the reference's 0 / 0)."""
    import variantformer_amd.seq2reg.model as s2r
    from variantformer_amd import ops
    from variantformer_amd.seq2gene.modules.layers import ln_fold_forced_off
    import contextlib
    torch.manual_seed(5)
    m = s2r.Seq2RegPredictor(vocab_size=500, embedding_dim=512, num_heads=8, num_layers=2, num_tissues=2, num_classes=2,
                             token_length=200, use_flash=True, positional_encoding="sinusoidal", seq_pool="mean").cuda()
    with torch.no_grad():
        for prm in m.parameters():
            if prm.dim() > 1:
                prm.mul_(1.5)
    rng = np.random.default_rng(3)
    lens = [200, 1, 2, 63, 64, 65, 199, 0] + list(rng.integers(1, 201, 56))
    W, L = len(lens), 200
    ids = torch.from_numpy(rng.integers(1, 500, (W, L))).long()
    pad = torch.ones((W, L), dtype=torch.bool)
    for w, n in enumerate(lens):
        pad[w, :n] = False
    from variantformer_amd import runtime
    outs = {}
    for flag in (True, False):
        with runtime.override(pool_before_down_projection=flag), \
                ops.compute_dtype(torch.bfloat16 if precision == "bf16-mixed" else torch.float16), \
                (contextlib.nullcontext() if ln_fold else ln_fold_forced_off()), torch.no_grad():
            outs[flag] = m.embed_packed(ids.cuda(), pad.cuda(), int((~pad).sum()), torch.float32).cpu()
    new, ref = outs[True], outs[False]
    assert torch.isnan(new[7]).all() and torch.isnan(ref[7]).all()
    keep = [w for w in range(W) if w != 7]
    scale = float(ref[keep].abs().max())
    assert scale > 0.5
    assert float((new[keep] - ref[keep]).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("precision", ["bf16-mixed", "16-mixed"])
@pytest.mark.parametrize("pos", ["sinusoidal", "alibi"])
def test_seq2reg_first_layer_qkv_lookup_is_bit_identical(precision, pos):
    """The encoder's first layer looks norm1 -> Wqkv up per distinct input row (Embedding(id) + positional(position): vocab x
    token_length rows, Seq2RegPredictor._layer0_qkv_table + vf_token_keys + a row gather) instead of projecting every token.  A
    GEMM row depends on its own input row only and the table rows go through the same kernels: the pooled embeddings are
    BIT-identical to the per-token projection, for both operand types, with the positional table (key = id * L + position) and
    without (ALiBi: key = id); ragged windows incl. interior pads, an empty window, ids at both ends of the vocabulary, and ids
    outside it (clamped like nn.Embedding's kernel clamps them)."""
    import variantformer_amd.seq2reg.model as s2r
    from variantformer_amd import ops
    torch.manual_seed(11)
    m = s2r.Seq2RegPredictor(vocab_size=500, embedding_dim=512, num_heads=8, num_layers=2, num_tissues=2, num_classes=2,
                             token_length=200, use_flash=True, positional_encoding=pos, seq_pool="mean").cuda()
    rng = np.random.default_rng(4)
    lens = [200, 1, 2, 63, 64, 65, 199, 0] + list(rng.integers(1, 201, 140))     # 148 windows x 8 heads: the row-map attention kernel serves it
    W, L = len(lens), 200
    ids = torch.from_numpy(rng.integers(0, 500, (W, L))).long()
    ids[0, :4] = torch.tensor([0, 499, 700, -3])
    pad = torch.ones((W, L), dtype=torch.bool)
    for w, n in enumerate(lens):
        pad[w, :n] = False
    pad[6, 50:60] = True                                             # pads inside a window: positions are not ranks
    outs = {}
    from variantformer_amd import runtime
    assert ops.attn_rows_supported(64, False, W, 8, 200, 200, True)
    for flag in (True, False, "gathered"):        # lookup + gather in the attention's loads / no lookup / lookup + row gather
        with runtime.override(layer0_qkv_table=bool(flag), rows_in_attention=flag is True), \
                ops.compute_dtype(torch.bfloat16 if precision == "bf16-mixed" else torch.float16), torch.no_grad():
            outs[flag] = m.embed_packed(ids.cuda(), pad.cuda(), int((~pad).sum()), torch.float32).cpu()
    keep = [w for w in range(W) if w != 7]
    assert torch.isfinite(outs[False][keep]).all() and float(outs[False][keep].abs().max()) > 0.1
    assert torch.equal(outs[True][keep], outs[False][keep]) and torch.equal(outs["gathered"][keep], outs[False][keep])
    assert torch.isnan(outs[True][7]).all()
    # the keys themselves
    cu = ops.mask_to_cu_seqlens(pad.cuda())
    keys = ops.token_keys(ids.cuda(), pad.cuda(), cu, int((~pad).sum()), 500, 200).cpu()
    cl = ids.clamp(0, 499)
    expect = torch.cat([(cl[w] * 200 + torch.arange(L))[~pad[w]] for w in range(W)])
    assert torch.equal(keys, expect)


def test_reference_signature_modulator_forward_padded(golden):
    """CombinedModulator.forward with the reference's padded / per-tissue-repeated arguments reproduces the
    fixture's padded gene output (zeros at padded positions)."""
    meta, arrays, sd, batch = golden
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    model.trainer = None
    X, mask, labels, precision, donors = model.transform_with_batching(
        batch["cre_sequences"], batch["cre_attention_masks"], batch["tissue_context"], batch["ref_cre_labels"],
        batch["strand_val"], embedder=model.cre_tokenizer)
    assert precision == torch.float32 and donors == list(range(len(meta["n_cres"])))
    assert X.shape[:2] == (len(meta["n_cres"]), max(meta["n_cres"])) and mask.dtype == torch.bool
    Xg, maskg, _, _, _ = model.transform_with_batching(
        batch["gene_embeddings"], batch["gene_attention_masks"], batch["tissue_context"],
        [torch.zeros(len(g)) for g in batch["gene_embeddings"]], batch["strand_val"], embedder=model.gene_tokenizer)
    from variantformer_amd import ops
    from variantformer_amd.seq2gene.modules.layers import packed_linear
    w, b = packed_linear(model.cre_map)
    x = ops.gemm(ops.cast_bf16(X.view(-1, X.shape[-1]).contiguous()), w, b, ops.EPI_F32).view(*X.shape[:2], -1)
    w, b = packed_linear(model.gene_map)
    xg = ops.gemm(ops.cast_bf16(Xg.view(-1, Xg.shape[-1]).contiguous()), w, b, ops.EPI_F32).view(*Xg.shape[:2], -1)
    reps = torch.tensor([len(t) for t in meta["tissues"]], device="cuda")
    tv = torch.tensor([t for ts in meta["tissues"] for t in ts], device="cuda").unsqueeze(1)
    x, mask, labels = (torch.repeat_interleave(t, reps, 0) for t in (x, mask, labels))
    xg, maskg = torch.repeat_interleave(xg, reps, 0), torch.repeat_interleave(maskg, reps, 0)
    g, _, maskg = model.combined_modulator.prepare_input(xg, "multi_registry", model.start_tkn, tv, maskg)
    out, gt, ct = model.combined_modulator(x, g, context=labels, cre_padding_mask=mask, gene_padding_mask=maskg,
                                           context_padding_mask=mask)
    ref = arrays["modulator_gene_out"]
    assert out.shape == ref.shape and _rel(out.cpu().numpy(), ref) < BF16_VS_FP32
    assert float(out[maskg].abs().sum()) == 0.0                       # pad_input zero fill
    assert float(gt.abs().sum()) == 0.0 and float(ct.abs().sum()) == 0.0


@pytest.mark.parametrize("case", ["prod_dims_small", "prod_dims_ragged"])
def test_production_dims_vs_oracle(case):
    """Production widths (seq2reg d=512/h=8, modulator D=1536/H=32/dh=48, 200-token windows), fewer layers and
    genes so the CPU oracle finishes in seconds; seeded weights."""
    if case == "prod_dims_small":
        layers, n_cres, n_chunks, tissues = 2, [12], [5], [TISSUES_54[:3]]
    else:
        layers, n_cres, n_chunks, tissues = 3, [7, 40, 1], [3, 9, 2], [[7], TISSUES_54[:5], [62, 10]]
    kw = seq2gene_kw(layers=layers)
    model = build_model(SEQ2REG_512, kw, seed=4242)
    sd = state_dict_cpu(model)
    model = model.cuda()
    batch = make_batch(99, n_cres, n_chunks, tissues, 200)
    out = model.predict_step(batch, 3)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    ghp = O.Seq2GeneHP.from_kwargs(kw)
    orc = O.predict_step(batch, sd, hp, hp, ghp, rounding="bf16", share_cre_stream=True)
    for i in range(len(n_cres)):
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(out["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL
    check_signal(case, out["pred_gene_exp"], orc["pred_gene_exp"])


def test_separate_layernorm_pass_mode_vs_oracle(monkeypatch):
    """VF_LN_FOLD=0: LayerNorm as a kernel of its own (vf_layernorm) in front of plain GEMMs -- the rounding points of
    round 1 -- against the oracle with fold_ln=False; and the default (LayerNorm folded into the GEMM epilogues) differs
    from it only at bf16 level on the same inputs."""
    monkeypatch.delenv("VF_LN_FOLD", raising=False)                    # the first run is the default contract whatever the ambient switch
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=4242)
    sd = state_dict_cpu(model)
    model = model.cuda()
    batch = make_batch(99, [12, 5], [5, 3], [TISSUES_54[:3], [9]], 200)
    folded = model.predict_step(batch, 0)
    monkeypatch.setenv("VF_LN_FOLD", "0")
    plain = model.predict_step(batch, 0)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    ghp = O.Seq2GeneHP.from_kwargs(kw)
    orc = O.predict_step(batch, sd, hp, hp, ghp, rounding=O.Rounding("bf16", fold_ln=False), share_cre_stream=True)
    for i in range(2):
        assert prel(plain["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(plain["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL
        assert prel(plain["pred_gene_exp"][i], folded["pred_gene_exp"][i]) < 10 * NORTH_STAR_RTOL
        assert not np.array_equal(plain["embeddings"][i], folded["embeddings"][i])      # two different rounding contracts
    check_signal("VF_LN_FOLD=0 vs oracle(fold_ln=False)", plain["pred_gene_exp"], orc["pred_gene_exp"])


def test_first_gene_layer_projection_dedup_is_exact(monkeypatch):
    """Gene layer 0 projects LayerNorm1 -> Wqkv once per distinct row of the gene stream (chunk rows are the same for
    every tissue there) and expands the result: bit-identical to projecting every row of the expanded stream."""
    from variantformer_amd.seq2gene.modules.layers import ContextFlashAttentionEncoderLayer
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=77).cuda()
    batch = make_batch(5, [30, 11], [4, 6], [TISSUES_54[:7], [9, 33, 2]], 200)
    a = model.predict_step(batch, 0)
    monkeypatch.setattr(ContextFlashAttentionEncoderLayer, "self_qkv_of_unique_rows", lambda self, *args, **kw: None)
    b = model.predict_step(batch, 0)
    for i in range(2):
        np.testing.assert_array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
        np.testing.assert_array_equal(a["embeddings"][i], b["embeddings"][i])


def test_first_gene_layer_input_from_16bit_copies_of_the_distinct_rows_is_exact(monkeypatch):
    """The stream entering gene layer 0 is consumed only through its 16-bit operand copy (the self-attention block's residual)
    and its fp16 trunk copy (the down-projection's residual): both are gathered from the copies of the DISTINCT rows, and the
    fp32 [sum T x G, D] stream is never built (bf16 operands; fp16 operands keep the fp32 gather).  Casts are row-wise:
    bit-identical to gathering fp32 rows first."""
    import variantformer_amd.seq2gene.modules.layers as Lyr
    monkeypatch.delenv("VF_LN_FOLD", raising=False)                    # the form under test belongs to the default contract
    monkeypatch.delenv("VF_TRUNK16", raising=False)
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=79).cuda()
    batch = make_batch(7, [9, 5], [150, 20], [TISSUES_54[:3], [9, 33]], 200)
    orig = Lyr.ContextFlashAttentionEncoderLayer.self_qkv_of_unique_rows
    calls = {"stream": 0}

    def no_stream(self, *a, with_stream=False, **k):             # the round-4 form: projection lookup only, fp32 stream gathered
        r = orig(self, *a, with_stream=with_stream, **k)
        if with_stream and r is not None:
            calls["stream"] += r[1] is not None
            return r[0], None
        return r
    for precision in ("bf16-mixed", "16-mixed"):
        model.trainer = type("T", (), {"precision": precision})()
        a = model.predict_step(batch, 0)
        monkeypatch.setattr(Lyr.ContextFlashAttentionEncoderLayer, "self_qkv_of_unique_rows", no_stream)
        b = model.predict_step(batch, 0)
        monkeypatch.setattr(Lyr.ContextFlashAttentionEncoderLayer, "self_qkv_of_unique_rows", orig)
        for i in range(2):
            np.testing.assert_array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
            np.testing.assert_array_equal(a["embeddings"][i], b["embeddings"][i])
    assert calls["stream"] == 1, "the 16-bit stream form serves bf16 operands (fp16 operands add fp32 rows in the first down-projection)"


def test_counted_context_keys_equal_the_expanded_context_attention(monkeypatch):
    """The CRE layers' context cross attention over the 9 distinct label rows with counts (vf_attn_counted_keys) against the
    round-4 form over the gathered [N, 2D] rows: the same function, different rounding points (no 16-bit P) -- expression and
    embeddings agree at the 16-bit level, both within the north-star bar of their own same-rounding oracle; also on the
    separate-LayerNorm path the self-healing recomputation takes."""
    from variantformer_amd import runtime
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=81)
    sd = state_dict_cpu(model)
    model = model.cuda()
    batch = make_batch(8, [40, 7, 1], [4, 3, 2], [TISSUES_54[:3], [9, 33], [12]], 200)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    ghp = O.Seq2GeneHP.from_kwargs(kw)
    for fold in ("1", "0"):
        monkeypatch.setenv("VF_LN_FOLD", fold)
        res = {}
        # "lowrank" (the default): two skinny GEMMs around a 9-way softmax; True: q -> counted-key attention -> out_proj;
        # False: the round-4 form over the gathered [N, 2D] rows
        for flag in ("lowrank", True, False):
            monkeypatch.setattr(O, "COUNTED_CONTEXT_KEYS", bool(flag))        # the oracle's own switches (test infrastructure)
            monkeypatch.setattr(O, "LOWRANK_CONTEXT", flag == "lowrank")
            with runtime.override(counted_context_keys=bool(flag), lowrank_context=flag == "lowrank"):
                out = model.predict_step(batch, 0)
            orc = O.predict_step(batch, sd, hp, hp, ghp, rounding=O.Rounding("bf16", fold_ln=fold == "1"), share_cre_stream=True)
            for i in range(3):
                assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
                assert _rel(out["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL
            res[flag] = out
        for i in range(3):
            assert prel(res[True]["pred_gene_exp"][i], res[False]["pred_gene_exp"][i]) < NORTH_STAR_RTOL
            assert prel(res["lowrank"]["pred_gene_exp"][i], res[False]["pred_gene_exp"][i]) < NORTH_STAR_RTOL
            assert not np.array_equal(res[True]["embeddings"][i], res[False]["embeddings"][i]) or i == 2   # (gene 2: one CRE, one label)
            assert not np.array_equal(res["lowrank"]["embeddings"][i], res[True]["embeddings"][i])


def test_first_gene_layer_row_map_attention_is_exact(monkeypatch):
    """With 129-256 gene tokens the first gene layer's self attention reads the distinct projected rows THROUGH the row map
    (vf_attn_varlen_fwd_rows) instead of a materialised [tokens, 3 D] gather: bit-identical to the gathered form and to
    projecting every row (ragged batch: 151- and 21-token sequences in one launch)."""
    import variantformer_amd.seq2gene.modules.layers as Lyr
    from variantformer_amd import ops
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=78).cuda()
    batch = make_batch(6, [9, 5], [150, 20], [TISSUES_54[:3], [9, 33]], 200)
    assert ops.attn_rows_supported(48, True, 5, 32, 151, 151, True)
    from variantformer_amd import runtime
    a = model.predict_step(batch, 0)
    with runtime.override(rows_in_attention=False):
        b = model.predict_step(batch, 0)
        monkeypatch.setattr(Lyr.ContextFlashAttentionEncoderLayer, "self_qkv_of_unique_rows", lambda self, *args, **kw: None)
        c = model.predict_step(batch, 0)
    for i in range(2):
        np.testing.assert_array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
        np.testing.assert_array_equal(a["embeddings"][i], b["embeddings"][i])
        np.testing.assert_array_equal(a["embeddings"][i], c["embeddings"][i])


def test_cre_stream_on_a_side_stream_is_bit_identical(monkeypatch):
    """runtime.Switches.overlap_cre_stream (the product default): the CRE layers on a side stream beside the gene layers (CRE
    layer i + 1 needs CRE layer i only; gene layer i + 1 needs gene layer i and CRE layer i).  The overlapped path is
    bit-identical to the single-stream one over repeated calls (the first forward of a configuration runs single-stream by
    itself: it builds the per-weights caches), a LayerNorm-fold alert raised by a CRE-stream kernel on the side stream still
    reaches the batch (the model recomputes it), and stale bits on the side stream's flag do not (round-5 advice).  Full
    depth: test_default_path_is_reproducible_at_full_depth."""
    import variantformer_amd.seq2gene.model_combined_modulator as M
    from variantformer_amd import ops, runtime
    monkeypatch.delenv("VF_LN_FOLD", raising=False)                    # the alert half needs the fold (default contract)
    monkeypatch.delenv("VF_TRUNK16", raising=False)
    assert runtime.switches().overlap_cre_stream, "the product default runs the CRE layers on the side stream"
    model = build_model(SEQ2REG_512, seq2gene_kw(layers=5), seed=3).cuda()
    batch = make_batch(4, [300, 40], [150, 20], [TISSUES_54[:5], [9, 33]], 200)
    with runtime.override(overlap_cre_stream=False):
        a = model.predict_step(batch, 0)
    warm = model.combined_modulator.cre_layers[0].__dict__["_vf_overlap_warm"]
    assert len(warm) == 1
    hs = M.heal_state(model)
    for _ in range(3):
        b = model.predict_step(batch, 0)
        for i in range(2):
            np.testing.assert_array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
            np.testing.assert_array_equal(a["embeddings"][i], b["embeddings"][i])
    assert hs.batches == 0 and hs.finished == 4
    # stale bits on the side stream's flag belong to nobody: the next batch must not be recomputed because of them
    dev = torch.device("cuda", torch.cuda.current_device())
    side = M._side_stream(dev)
    with torch.cuda.stream(side):
        ops._alert_flag(dev).fill_(1)
    torch.cuda.synchronize()
    b = model.predict_step(batch, 0)
    assert hs.batches == 0, "a stale bit on the side stream's flag was taken for this batch's alert"
    np.testing.assert_array_equal(a["embeddings"][0], b["embeddings"][0])
    # an alert that only a CRE-stream kernel can raise: a huge common offset on the output bias of CRE layer 1's down-projection
    # gives every row of the CRE stream a mean of many standard deviations from CRE layer 2 on; the gene stream never sees it in
    # a LayerNorm-folded statistic of its own (it reads the CRE stream through K / V projections only)
    monkeypatch.setattr(M, "LN_HEAL_STICKY_AFTER", 10 ** 9)
    with torch.no_grad():
        model.combined_modulator.cre_layers[1].linear_geglu_2.bias += 60.0
    with runtime.override(overlap_cre_stream=False):
        want = model.predict_step(batch, 0)
    assert hs.batches == 1, "the serial path must have flagged and recomputed the batch"
    got = model.predict_step(batch, 0)
    assert hs.batches == 2, "the alert raised on the side stream must reach the batch"
    for i in range(2):
        np.testing.assert_array_equal(got["pred_gene_exp"][i], want["pred_gene_exp"][i])


def test_default_path_is_reproducible_at_full_depth():
    """The product path at full depth (25 modulator layers, 6 seq2reg layers; CRE layers on the side stream): the same ragged
    batch evaluated three times gives the same bits -- the bits of the single-stream order -- and once more inside a different
    batch the same values.  (Until round 6 two streams differed run to run by 7e-4: packed-fp32 instructions beside another
    kernel's MFMAs, tests/test_concurrency_gpu.py.)"""
    import bench
    from variantformer_amd import runtime
    assert runtime.switches().overlap_cre_stream
    model, hp, kw = bench.build_model(torch.device("cuda", 0))
    batch = make_batch(77, [700, 90, 311], [150, 20, 64], [TISSUES_54, TISSUES_54[:7], TISSUES_54[:30]], 200)
    a = model.predict_step(batch, 0)                               # (the first forward builds the caches on one stream)
    for _ in range(3):
        b = model.predict_step(batch, 0)
        for i in range(3):
            np.testing.assert_array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
            np.testing.assert_array_equal(a["embeddings"][i], b["embeddings"][i])
    with runtime.override(overlap_cre_stream=False):
        s1 = model.predict_step(batch, 0)
    for i in range(3):
        np.testing.assert_array_equal(s1["pred_gene_exp"][i], b["pred_gene_exp"][i])
        np.testing.assert_array_equal(s1["embeddings"][i], b["embeddings"][i])
    two = {k: v[1:] for k, v in batch.items()}
    c = model.predict_step(two, 0)
    for i in range(2):
        np.testing.assert_allclose(c["pred_gene_exp"][i], a["pred_gene_exp"][i + 1], rtol=1e-5, atol=1e-6)


def test_tissue_invariance_and_batch_independence():
    """Size-independent properties: a gene's prediction does not depend on which other genes share its batch,
    nor on how many tissues are requested with it (the exact de-duplication must not leak between rows)."""
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=7).cuda()
    b_all = make_batch(5, [30, 11], [4, 6], [TISSUES_54[:4], [9, 33]], 200)
    one = {k: (v[:1] if isinstance(v, list) else v[:1]) for k, v in b_all.items()}
    a = model.predict_step(b_all, 0)
    b = model.predict_step(one, 0)
    np.testing.assert_allclose(a["pred_gene_exp"][0], b["pred_gene_exp"][0], rtol=1e-5, atol=1e-6)
    one_t = dict(one)
    one_t["tissue_context"] = [one["tissue_context"][0][2:3]]
    c = model.predict_step(one_t, 0)
    np.testing.assert_allclose(c["pred_gene_exp"][0], b["pred_gene_exp"][0][2:3], rtol=1e-5, atol=1e-6)


def test_variant_prediction_contract():
    """VEP wrapper (reference model_combined_modulator.py:909-1004): ref/het/hom batch with token positions."""
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=11).cuda()
    batch = make_batch(8, [6, 6, 6], [3, 3, 3], [[7, 8]] * 3, 200)
    vb = {"cre_sequences": batch["cre_sequences"], "cre_attention_masks": batch["cre_attention_masks"],
          "tissue_context": batch["tissue_context"], "ref_labels": batch["ref_cre_labels"], "strand": batch["strand_val"],
          "gene_embeddings": batch["gene_embeddings"], "gene_attention_masks": batch["gene_attention_masks"],
          "cre_token_position": torch.tensor([2.0, 2.0, 2.0]), "gene_token_position": torch.tensor([1.0, 1.0, 1.0]),
          "variant_type": ["ref", "het", "hom"]}
    model.vep = True
    out = model.predict_step(vb, 0)
    assert set(out) == {"pred_gene_exp", "embd", "variant_type", "gene_token_embedding", "cre_token_embedding"}
    assert len(out["pred_gene_exp"]) == 3 and out["gene_token_embedding"][0].shape == (2, 1536)
    # CRE stream is tissue independent: both tissue rows carry the same CRE token embedding
    np.testing.assert_array_equal(out["cre_token_embedding"][1][0], out["cre_token_embedding"][1][1])
    assert np.abs(out["gene_token_embedding"][0]).sum() > 0
    model.vep = False
    plain = model.predict_step(batch, 0)
    np.testing.assert_allclose(out["pred_gene_exp"][2], plain["pred_gene_exp"][2], rtol=1e-6)


def test_full_depth_production_model_vs_oracle():
    """The full architecture (25 modulator layers, D=1536/H=32, seq2reg d=512/h=8/6 layers; 1.2 B random-init
    parameters as in bench.py) on a small gene, HIP vs the same-rounding oracle: error accumulated over all
    49 + 6 layers stays inside the north-star tolerance."""
    import bench
    model, hp, kw = bench.build_model(torch.device("cuda", 0))
    sd = state_dict_cpu(model)
    batch = make_batch(4321, [24, 9], [6, 3], [TISSUES_54[:3], [62]], 200)
    out = model.predict_step(batch, 0)
    shp = O.Seq2RegHP.from_hparams(hp)
    torch.set_num_threads(min(16, bench.host_threads()))
    orc = O.predict_step(batch, sd, shp, shp, O.Seq2GeneHP.from_kwargs(kw), rounding="bf16", share_cre_stream=True)
    f32 = O.predict_step(batch, sd, shp, shp, O.Seq2GeneHP.from_kwargs(kw), rounding=None, share_cre_stream=True)
    for i in range(2):
        assert np.isfinite(out["pred_gene_exp"][i]).all()
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(out["embeddings"][i], orc["embeddings"][i]) < 5e-3
        # the real bf16-operand deviation from fp32 arithmetic at full depth, element-wise (|a-b| / (|b| + rms(b)))
        ep, ee = prel(out["pred_gene_exp"][i], f32["pred_gene_exp"][i]), _erel(out["embeddings"][i], f32["embeddings"][i])
        print(f"[full depth] gene {i}: HIP vs fp32 oracle element-wise: expression {ep:.2e}, embedding {ee:.2e}; "
              f"vs same-rounding oracle: expression {_erel(out['pred_gene_exp'][i], orc['pred_gene_exp'][i]):.2e}, "
              f"embedding {_erel(out['embeddings'][i], orc['embeddings'][i]):.2e}")
        # bf16 OPERANDS against pure fp32 ARITHMETIC at full depth: the expression still meets the north-star bar; the
        # 1536-wide embedding rows carry the bf16 noise of 49 layers (measured 7e-3 ... 1.1e-2 element-wise)
        assert ep < NORTH_STAR_RTOL and ee < 2e-2
    check_signal("full depth vs same-rounding oracle", out["pred_gene_exp"], orc["pred_gene_exp"])
    check_signal("full depth, bf16 operands vs pure fp32 oracle", out["pred_gene_exp"], f32["pred_gene_exp"])


def test_vep_window_dedupe_is_exact():
    """ref / het / hom batches share every window but the one carrying the variant: embedding unique windows once
     must give bit-identical results to embedding all of them."""
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=21).cuda()
    one = make_batch(77, [9], [4], [[7, 8, 9]], 200)
    batch = {k: (v * 3 if isinstance(v, list) else v.repeat(3, 1)) for k, v in one.items()}
    batch["cre_sequences"] = [t.clone() for t in batch["cre_sequences"]]
    batch["gene_embeddings"] = [t.clone() for t in batch["gene_embeddings"]]
    batch["cre_sequences"][1][4, 0, 10] = 17          # het: one token changes in CRE window 4
    batch["cre_sequences"][2][4, 0, 10] = 6           # hom
    batch["gene_embeddings"][2][1, 0, 3] = 9
    with torch.no_grad():
        pa = model.prepare_batch(batch, dedupe_windows=False)
        pd = model.prepare_batch(batch)                 # the default: exact de-duplication whenever it removes windows
        assert pd.windows_total == (27, 12) and pd.windows_embedded == (11, 5)
        assert pd.cre_ids.shape[0] == 9 + 2 and pd.gene_ids.shape[0] == 4 + 1 and pa.cre_ids.shape[0] == 27
        a = model.forward_prepared(pa)
        d = model.forward_prepared(pd)
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])


def test_cross_gene_window_dedupe_on_the_normal_predict_path():
    """Neighbouring genes of a whole-genome scan share cCRE windows byte for byte (reference datasets/vcfdataset.py:219-283)
    while their gene-body chunks differ: prepare_batch's default de-duplication embeds every distinct window once --
    CRE windows and gene chunks independently -- and the outputs are bit-identical to the un-shared evaluation."""
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=33).cuda()
    batch = make_batch(123, [9, 7, 5], [4, 3, 2], [[7, 8], [9], [10, 11, 12]], 200)
    for k in ("cre_sequences", "cre_attention_masks"):          # genes 1 and 2 reuse windows of gene 0 (and one of their own)
        batch[k][1][2:6] = batch[k][0][3:7]
        batch[k][2][0:3] = batch[k][0][6:9]
        batch[k][2][4] = batch[k][2][3]
    with torch.no_grad():
        pa = model.prepare_batch(batch, dedupe_windows=False)
        pd = model.prepare_batch(batch)
        assert pd.windows_total == (21, 9) and pd.windows_embedded == (13, 9)
        assert pd.gene_unique_inverse is None and pd.cre_unique_inverse is not None
        a = model.forward_prepared(pa)
        d = model.forward_prepared(pd)
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])
    out = model.predict_step(batch, 0)                          # the product entry point takes the shared path
    assert np.array_equal(np.concatenate(out["pred_gene_exp"]), a[0].float().cpu().numpy())


@pytest.mark.parametrize("name", ["small_twomod", "small_twomod_b", "small_twomod_c"])
def test_two_module_variant_vs_reference_golden(name):
    """model_class Seq2GenePredictor: same network under the older module layout; outputs vs the reference's own
    Seq2GenePredictor -- shipped options (small_twomod), the class's default cross-attention-only gene layers with
    cross_alibi (small_twomod_b), context-free CRE layers (small_twomod_c)."""
    from variantformer_amd.seq2gene.model import Seq2GenePredictor
    from variantformer_amd.seq2reg.model import Seq2RegPredictor
    meta, arrays, sd, batch = load_fixture(name)
    m = Seq2GenePredictor(cre_tokenizer=Seq2RegPredictor(**meta["seq2reg"]), gene_tokenizer=Seq2RegPredictor(**meta["seq2reg"]),
                          **meta["seq2gene"])
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    b = dict(batch)
    b["cre_attention_mask"] = b.pop("cre_attention_masks")           # the key this class reads in the reference
    out = m.predict_step(b, 0)
    for i in range(len(meta["n_cres"])):
        assert prel(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"]) < NORTH_STAR_RTOL
        assert _rel(out["embeddings"][i], arrays[f"embeddings_{i}"]) < 5e-3
    check_signal(f"two-module class ({name}) vs reference fp32", out["pred_gene_exp"],
                 [arrays[f"pred_gene_exp_{i}"] for i in range(len(meta["n_cres"]))])


def test_edge_geometries_vs_oracle():
    """Extremes of SURVEY §8d cfg 3 in one batch: a gene with a single CRE and a single chunk and one tissue, a gene
    with 2048 CREs (clip) and 20 chunks, a gene with 200 full chunks and 54 tissues; production widths, 2 layers."""
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=99)
    sd = state_dict_cpu(model)
    model = model.cuda()
    batch = make_batch(2025, [1, 2048, 40], [1, 20, 200], [[62], [7, 30], TISSUES_54], 200)
    out = model.predict_step(batch, 0)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    torch.set_num_threads(8)
    orc = O.predict_step(batch, sd, hp, hp, O.Seq2GeneHP.from_kwargs(kw), rounding="bf16", share_cre_stream=True)
    for i in range(3):
        assert out["pred_gene_exp"][i].shape == orc["pred_gene_exp"][i].shape
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(out["embeddings"][i], orc["embeddings"][i]) < 5e-3
    check_signal("edge geometries", out["pred_gene_exp"], orc["pred_gene_exp"])


def test_last_layer_registry_only_path_is_exact():
    """forward_prepared computes only the registry rows of the last gene layer; the full-layer path (used by VEP)
    must give the same embeddings / predictions."""
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=5).cuda()
    batch = make_batch(31, [20, 7], [5, 9], [TISSUES_54[:4], [8, 62]], 200)
    with torch.no_grad():
        pb = model.prepare_batch(batch)
        fast = model.forward_prepared(pb)
        full = model.forward_prepared(pb, return_cre=True)
    np.testing.assert_allclose(fast[1].cpu().numpy(), full[1].cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(fast[0].cpu().numpy(), full[0].cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_headline_size_gene_vs_oracle_and_properties():
    """BASELINE configs[1] geometry at full size (N=1024 cCRE windows, C=200 gene chunks, T=54 tissues, 25 modulator
    layers, seq2reg 6 layers): the HIP path against the same-rounding CPU oracle on one whole gene, plus the
    size-independent properties at that size -- a gene's result must not depend on the genes sharing its batch, and a
    tissue's result must not depend on the other tissues requested."""
    import bench
    model, hp, kw = bench.build_model(torch.device("cuda", 0))
    sd = state_dict_cpu(model)
    batch = make_batch(20251205, [1024, 700], [200, 150], [TISSUES_54, TISSUES_54[:20]], 200)
    both = model.predict_step(batch, 0)
    assert both["pred_gene_exp"][0].shape == (54, 1) and both["pred_gene_exp"][1].shape == (20, 1)
    assert all(np.isfinite(p).all() and (p >= 0).all() for p in both["pred_gene_exp"])      # Softplus output
    first = {k: v[:1] for k, v in batch.items()}
    alone = model.predict_step(first, 0)
    np.testing.assert_allclose(alone["pred_gene_exp"][0], both["pred_gene_exp"][0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alone["embeddings"][0], both["embeddings"][0], rtol=1e-5, atol=1e-6)
    few = dict(first)
    few["tissue_context"] = [first["tissue_context"][0][[3, 30, 53]]]
    sub = model.predict_step(few, 0)
    np.testing.assert_allclose(sub["pred_gene_exp"][0], alone["pred_gene_exp"][0][[3, 30, 53]], rtol=1e-5, atol=1e-6)
    # whole-gene parity against the oracle (executes the de-duplicated ~18 TFLOP on the host: tens of seconds)
    shp = O.Seq2RegHP.from_hparams(hp)
    torch.set_num_threads(min(16, bench.host_threads()))
    orc = O.predict_step(first, sd, shp, shp, O.Seq2GeneHP.from_kwargs(kw), rounding="bf16", share_cre_stream=True)
    assert prel(alone["pred_gene_exp"][0], orc["pred_gene_exp"][0]) < NORTH_STAR_RTOL
    assert _rel(alone["embeddings"][0], orc["embeddings"][0]) < 5e-3
    check_signal("headline gene, 54 tissues", alone["pred_gene_exp"][0], orc["pred_gene_exp"][0])
    # The literal north-star bar at the headline size (round-5 verdict: the same-rounding oracle restates the kernels' rounding
    # points, so at full size it is a consistency check): the SAME gene through the oracle in PURE fp32 arithmetic
    # (rounding=None: no 16-bit rounding anywhere, plain LayerNorm, the reference's formulation), against the HIP path with bf16
    # operands (the shipped bf16-mixed) and with fp16 operands (16-mixed).  Expression within 1e-3 relative, element-wise; the
    # embeddings' element-wise error is printed and bounded at 1.5x what MI355X measured (profiles/r06_*: the bf16 noise of 49
    # layers, no worse than the reference's own autocast -- tests/test_trained_like_gpu.py -- but 7-20x the expression's bar;
    # DESIGN.md section 5 has the table).
    pure = O.predict_step(first, sd, shp, shp, O.Seq2GeneHP.from_kwargs(kw), rounding=None, share_cre_stream=True)
    model.precision = "16-mixed"
    alone16 = model.predict_step(first, 0)
    model.precision = None
    for tag, got, emb_max_lim, emb_el_lim in (("bf16", alone, HEADLINE_EMB_MAXNORM["bf16"], HEADLINE_EMB_ELEMENTWISE["bf16"]),
                                              ("fp16", alone16, HEADLINE_EMB_MAXNORM["fp16"], HEADLINE_EMB_ELEMENTWISE["fp16"])):
        e_expr = prel(got["pred_gene_exp"][0], pure["pred_gene_exp"][0])
        e_max = _rel(got["embeddings"][0], pure["embeddings"][0])
        e_el = _erel(got["embeddings"][0], pure["embeddings"][0])
        s_rel = check_signal(f"headline gene vs PURE fp32, {tag} operands", got["pred_gene_exp"][0], pure["pred_gene_exp"][0])
        print(f"[headline vs pure fp32, {tag} operands] expression prel {e_expr:.2e} (bar 1e-3), signal-relative {s_rel:.2e}; "
              f"embeddings max-norm {e_max:.2e} (limit {emb_max_lim:g}), element-wise {e_el:.2e} (limit {emb_el_lim:g})")
        assert e_expr < NORTH_STAR_RTOL, (tag, e_expr)
        assert e_max < emb_max_lim and e_el < emb_el_lim, (tag, e_max, e_el)


@pytest.mark.parametrize("name", ["small_opts_a", "small_opts_b", "small_opts_c", "small_opts_d"])
def test_non_shipped_options_vs_reference_golden(name):
    """Layer options the shipped configuration leaves off, pinned to fixtures produced by the
    reference's own classes: (a) cross-attention-only gene layers + gene residual + ALiBi on the gene->CRE cross
    attention + start-token pooling, (b) tissue embedding added to the CRE tokens + max pooling, (c) one small MLP head
    per tissue, (d) context-free CRE layers + a shared linear head."""
    meta, arrays, sd, batch = load_fixture(name)
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    general = name in ("small_opts_a", "small_opts_b")
    assert model._general == general
    out = model.predict_step(batch, 0)
    cre_hp, gene_hp, hp = _hps(meta)
    orc = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding="bf16", share_cre_stream=not general)
    for i in range(len(meta["n_cres"])):
        p, e = out["pred_gene_exp"][i], out["embeddings"][i]
        assert p.shape == (len(meta["tissues"][i]), 1) and e.shape == (len(meta["tissues"][i]), meta["seq2gene"]["emb_dim"])
        emb_tol = 1e-2 if meta["seq2gene"]["gene_pooling"] == "max" else 5e-3   # max pooling keeps per-column extremes
        assert prel(p, arrays[f"pred_gene_exp_{i}"]) < NORTH_STAR_RTOL                # of bf16-noisy rows
        assert _rel(e, arrays[f"embeddings_{i}"]) < emb_tol
        assert prel(p, orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _rel(e, orc["embeddings"][i]) < emb_tol
    # Signal-relative bound for these 96-wide, 2-layer option fixtures: 5e-2.  Their across-tissue spread is tiny and the
    # statistic is the MAXIMUM over a handful of outputs, so it samples the 16-bit rounding noise rather than measuring it:
    # small_opts_b (max pooling) reads 2.0e-2 with the unscaled query projection, 3.3e-2 with the softmax scale folded into
    # the query weights (VF_Q_PRESCALE, the default) -- while the production-width, full-depth model, measured over three
    # geometries with either setting, sits at 0.7 ... 1.5e-2 and is closer to pure fp32 WITH the folding in all three
    # (profiles/r03_m_q_prescale_accuracy.log).  The suite-wide bound (helpers.SIGNAL_RTOL = 2e-2 since round 6; measured maximum 1.9e-2: cfg3) is kept everywhere else.
    check_signal(name + " vs reference fp32", out["pred_gene_exp"],
                 [arrays[f"pred_gene_exp_{i}"] for i in range(len(meta["n_cres"]))], tol=5e-2)
    if not general:
        return
    # the reference-signature forward with return_embedding (the VEP call) works for these option sets too; its
    # token-position outputs are pinned by tests/test_configs_gpu.py::test_variant_prediction_option_sets_vs_reference_golden
    pred, donors, emb, gtok, ctok = model.forward(
        batch["cre_sequences"], batch["cre_attention_masks"], batch["tissue_context"], batch["ref_cre_labels"],
        batch["strand_val"], batch["gene_embeddings"], batch["gene_attention_masks"], return_embedding=True)
    rows = sum(len(t) for t in meta["tissues"])
    assert pred.shape == (rows, 1) and emb.shape == gtok.shape == ctok.shape == (rows, meta["seq2gene"]["emb_dim"])
    assert float(gtok.abs().max()) == 0.0 and float(ctok.abs().max()) == 0.0      # no positions given (reference :296-326)
    np.testing.assert_allclose(emb.cpu().numpy(), np.concatenate(out["embeddings"]), rtol=1e-5, atol=1e-6)


def test_run_to_run_determinism_and_sequence_permutation():
    """No atomics and no data-dependent scheduling on the path: the same batch gives bit-identical results twice, and
    permuting the genes of a batch permutes the results (every kernel treats sequences independently)."""
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=21).cuda()
    batch = make_batch(31, [40, 7, 19], [5, 2, 9], [TISSUES_54[:4], [9], [33, 62]], 200)
    a = model.predict_step(batch, 0)
    b = model.predict_step(batch, 0)
    for i in range(3):
        assert np.array_equal(a["pred_gene_exp"][i], b["pred_gene_exp"][i])
        assert np.array_equal(a["embeddings"][i], b["embeddings"][i])
    perm = [2, 0, 1]
    pb = {k: ([v[j] for j in perm] if isinstance(v, list) else v[perm]) for k, v in batch.items()}
    c = model.predict_step(pb, 0)
    for new, old in enumerate(perm):
        np.testing.assert_allclose(c["pred_gene_exp"][new], a["pred_gene_exp"][old], rtol=1e-5, atol=1e-6)


def test_fp16_operand_mode_vs_oracle_and_reference_golden(golden):
    """precision "16-mixed" (BASELINE configs[4]: fp16 operands, fp32 accumulation; reference utils/functions.py:12-32,
    seq2gene/modules/layers.py:102-125): HIP vs the oracle with fp16 rounding points and vs the reference's fp32 fixture
    (fp16 carries 3 more mantissa bits than bf16, so it must sit closer to the fp32 run than the bf16 mode does)."""
    import types
    meta, arrays, sd, batch = golden
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    bf = model.predict_step(batch, 0)
    model.trainer = types.SimpleNamespace(precision="16-mixed")
    assert model.operand_dtype() == torch.float16
    out = model.predict_step(batch, 0)
    cre_hp, gene_hp, hp = _hps(meta)
    orc = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding="fp16", share_cre_stream=True)
    e16 = ebf = 0.0
    for i in range(len(meta["n_cres"])):
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _erel(out["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL
        assert prel(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"]) < NORTH_STAR_RTOL
        e16 = max(e16, _erel(out["embeddings"][i], arrays[f"embeddings_{i}"]))
        ebf = max(ebf, _erel(bf["embeddings"][i], arrays[f"embeddings_{i}"]))
    check_signal("fp16 operands vs oracle(fp16)", out["pred_gene_exp"], orc["pred_gene_exp"])
    print(f"[fp16 mode] embedding error vs the reference fp32 fixture: fp16 operands {e16:.2e}, bf16 operands {ebf:.2e}")
    assert e16 < ebf
    model.trainer = types.SimpleNamespace(precision="bf16-mixed")
    again = model.predict_step(batch, 0)
    assert np.array_equal(again["pred_gene_exp"][0], bf["pred_gene_exp"][0])     # weight repack follows the precision


def test_precision_32_mode_vs_reference_fixture():
    """trainer.precision = "32" on the HIP path (fp16 operands for every GEMM) against the reference's own precision-"32" run
    (MHA modules in fp16, everything else fp32: tests/golden/small_sin_p32.*): expression inside the north-star bar with a
    wide margin, embeddings at fp16 level."""
    import types
    meta, arrays, sd, batch = load_fixture("small_sin_p32")
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    model.trainer = types.SimpleNamespace(precision="32")
    assert model.operand_dtype() == torch.float16
    out = model.predict_step(batch, 0)
    for i in range(len(meta["n_cres"])):
        assert prel(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"]) < 2e-4
        assert _rel(out["embeddings"][i], arrays[f"embeddings_{i}"]) < 2e-3
    check_signal("precision 32 vs the reference's precision-32 run", out["pred_gene_exp"],
                 [arrays[f"pred_gene_exp_{i}"] for i in range(len(meta["n_cres"]))])


def test_fp16_operand_mode_production_width_vs_oracle():
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=4242)
    sd = state_dict_cpu(model)
    model = model.cuda()
    model.precision = "16-mixed"
    batch = make_batch(99, [7, 40, 1], [3, 9, 2], [[7], TISSUES_54[:5], [62, 10]], 200)
    out = model.predict_step(batch, 3)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    orc = O.predict_step(batch, sd, hp, hp, O.Seq2GeneHP.from_kwargs(kw), rounding="fp16", share_cre_stream=True)
    f32 = O.predict_step(batch, sd, hp, hp, O.Seq2GeneHP.from_kwargs(kw), rounding=None, share_cre_stream=True)
    for i in range(3):
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        assert _erel(out["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL
        assert prel(out["pred_gene_exp"][i], f32["pred_gene_exp"][i]) < NORTH_STAR_RTOL      # fp16 operands vs pure fp32
    check_signal("fp16 operands, production width vs oracle(fp16)", out["pred_gene_exp"], orc["pred_gene_exp"])
    check_signal("fp16 operands, production width vs pure fp32", out["pred_gene_exp"], f32["pred_gene_exp"])


def test_fp16_trunk_copy_default_vs_fp32_trunk_and_oracle(monkeypatch):
    """VF_TRUNK16=f16 (the default): between the layers of a stack the trunk exists as a scaled FP16 copy (11 significant
    bits) that the next down-projection adds as its residual (vf_gemm_ln_t16), not as fp32 rows; VF_TRUNK16=0 keeps fp32
    rows.  Production widths, 3 layers, ragged genes, bf16 operands: each mode within the north-star bar of the oracle
    with ITS rounding points (Rounding(trunk16="f16" / False)) and of pure fp32 arithmetic, and the fp16 copy costs no
    accuracy against fp32 that the operand roundings have not already spent (embedding error within 1.5x of the fp32
    trunk's; the bf16 trunk that round 3 also measured was ~4x)."""
    monkeypatch.delenv("VF_LN_FOLD", raising=False)                    # the trunk copy is part of the folded contract
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=515)
    sd = state_dict_cpu(model)
    model = model.cuda()
    batch = make_batch(31, [7, 40, 1], [3, 9, 2], [[7], TISSUES_54[:5], [62, 10]], 200)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    ghp = O.Seq2GeneHP.from_kwargs(kw)
    f32 = O.predict_step(batch, sd, hp, hp, ghp, rounding=None, share_cre_stream=True)
    outs, errs = {}, {}
    for mode, trunk in (("0", False), ("f16", "f16")):
        monkeypatch.setenv("VF_TRUNK16", mode)
        out = outs[mode] = model.predict_step(batch, 0)
        orc = O.predict_step(batch, sd, hp, hp, ghp, rounding=O.Rounding("bf16", trunk16=trunk), share_cre_stream=True)
        for i in range(3):
            assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL, (mode, i)
            assert prel(out["pred_gene_exp"][i], f32["pred_gene_exp"][i]) < NORTH_STAR_RTOL, (mode, i)
            assert _erel(out["embeddings"][i], orc["embeddings"][i]) < 3 * NORTH_STAR_RTOL, (mode, i)
        errs[mode] = [_erel(out["embeddings"][i], f32["embeddings"][i]) for i in range(3)]
        check_signal(f"trunk mode {mode} vs its oracle", out["pred_gene_exp"], orc["pred_gene_exp"])
    monkeypatch.delenv("VF_TRUNK16")
    dflt = model.predict_step(batch, 0)
    for i in range(3):
        assert np.array_equal(dflt["embeddings"][i], outs["f16"]["embeddings"][i]), "f16 must be the default mode"
        assert not np.array_equal(outs["0"]["embeddings"][i], outs["f16"]["embeddings"][i])
        print(f"[trunk] gene {i}: embedding vs pure fp32: fp32 trunk {errs['0'][i]:.2e}, fp16 copy {errs['f16'][i]:.2e}")
    assert max(errs["f16"]) < 1.5 * max(errs["0"])


def test_seq2reg_options_vs_reference_golden():
    """seq_pool max / linear, use_context (with and without expand_context), head dims 96 / 128 on the HIP path: vs the
    reference's own Seq2RegPredictor outputs (fp32 fixture) and vs the same-rounding oracle."""
    from tests.conftest import load_s2r_opts
    from variantformer_amd.seq2reg.model import Seq2RegPredictor
    for name, (hp, want, sd, g) in load_s2r_opts().items():
        m = Seq2RegPredictor(**hp)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        got = m(g["cre_sequences"], g["cre_attention_masks"], None, context=g["ref_cre_labels"], only_embed=True)
        assert got.shape == want.shape and got.dtype == torch.float32
        orc = O.seq2reg_embed(g["cre_sequences"], g["cre_attention_masks"], sd, "", O.Seq2RegHP.from_hparams(hp),
                              O.Rounding("bf16"), context=g["ref_cre_labels"])
        assert _erel(got.cpu().numpy(), orc.numpy()) < 1e-2, name            # bf16 flips on max-pooled / summed rows
        assert _erel(got.cpu().numpy(), want) < BF16_VS_FP32, name
    # a use_context tokenizer without integer labels fails like the reference's nn.Embedding does
    with pytest.raises(NotImplementedError, match="integer cCRE labels"):
        m_ctx = Seq2RegPredictor(**load_s2r_opts()["ctx_max"][0]).cuda()
        m_ctx(g["cre_sequences"], g["cre_attention_masks"], None, context=torch.zeros(9), only_embed=True)


def test_make_data_kv_layers_vs_reference_layer_fixture_and_oracle():
    """The layer option the models never switch on (make_data_kv: cross-attention queries from the raw context, keys / values
    from the normalised stream), for the three reference layer classes that carry it: HIP layers through the reference
    signature (padded + masks) against the reference's own layer outputs (tests/golden/layer_data_kv.*, fp32) and against the
    same-rounding oracle."""
    from tests.test_oracle_golden import _data_kv_fixture, data_kv_oracle
    from variantformer_amd.seq2gene.modules.layers import (ContextFlashAttentionEncoderLayer,
                                                           ContextFlashCrossAttentionEncoderLayer)
    z, meta, src, ctx, mask, keep, cu = _data_kv_fixture()
    D, H, F = meta["d_model"], meta["nhead"], meta["hidden_dim"]
    for name, info in meta["layers"].items():
        cls = ContextFlashCrossAttentionEncoderLayer if name == "cross" else ContextFlashAttentionEncoderLayer
        layer = cls(D, H, hidden_dim=F, dropout=0.0, use_alibi=info["use_alibi"], make_data_kv=True, mlp_dout=0.0)
        sd = {k[len(name) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + ".sd.")}
        layer.load_state_dict(sd, strict=True)
        layer = layer.cuda().eval()
        xs, cs = src[keep].cuda().contiguous(), ctx[keep].cuda().contiguous()
        with torch.no_grad():
            got = layer.forward_packed(xs, cu.cuda(), int((cu[1:] - cu[:-1]).max()), context=cs, cu_ctx=cu.cuda(),
                                       max_ctx=int((cu[1:] - cu[:-1]).max()))
            got = (got.x if hasattr(got, "x") else got).float().cpu().numpy()
            if name == "gene":          # the reference signature (padded tensors + masks) takes the same path
                pad = layer(src.cuda(), ctx.cuda(), src_key_padding_mask=mask.cuda(), precision=None)
                assert np.array_equal(pad.float().cpu().numpy()[keep.numpy()], got)
            orc = data_kv_oracle(name, sd, src[keep], ctx[keep], cu, H, info["use_alibi"], O.Rounding("bf16", fold_ln=False))
        want = z[f"{name}.out"][keep.numpy()]
        assert _rel(got, want) < BF16_VS_FP32, (name, _rel(got, want))
        assert _rel(got, orc.numpy()) < 5e-3, (name, _rel(got, orc.numpy()))
        print(f"[make_data_kv] {name}: vs reference fp32 {_rel(got, want):.2e}, vs same-rounding oracle {_rel(got, orc.numpy()):.2e}")
