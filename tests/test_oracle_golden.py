"""Pin the CPU oracle against vectors produced by the reference's own classes
(tests/golden/make_golden.py; fp32, CPU).  No GPU, no reference checkout needed."""
import json
import os

import numpy as np
import torch

from oracle import vf_oracle as O
from tests.conftest import GOLDEN

RTOL = 2e-5   # fp32 round-off between two fp32 CPU evaluations of the same graph
ATOL = 2e-5


def _hps(meta):
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    return hp, hp, O.Seq2GeneHP.from_kwargs(meta["seq2gene"])


def test_oracle_matches_reference_outputs(golden):
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, hp = _hps(meta)
    out = O.predict_step(batch, sd, cre_hp, gene_hp, hp)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(out["embeddings"][i], arrays[f"embeddings_{i}"], rtol=RTOL, atol=ATOL)
        assert out["pred_gene_exp"][i].shape == (len(meta["tissues"][i]), 1)
        assert out["embeddings"][i].dtype == np.float32


def test_oracle_matches_reference_intermediates(golden):
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, hp = _hps(meta)
    col = {}
    with torch.no_grad():
        O.forward(batch, sd, cre_hp, gene_hp, hp, collect=col)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(col["cre_tok"][i].numpy(), arrays[f"cre_tok_{i}"][:, 0], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(col["gene_tok"][i].numpy(), arrays[f"gene_tok_{i}"][:, 0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(col["first_gene_layer_out"].numpy(), arrays["first_gene_layer_out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(col["first_cre_layer_out"].numpy(), arrays["first_cre_layer_out"], rtol=RTOL, atol=ATOL)
    # padded [sum T, maxG, D] modulator output of the reference vs the per-gene packed oracle output
    ref = arrays["modulator_gene_out"]
    row = 0
    for i, t in enumerate(meta["tissues"]):
        o = col["modulator_gene_out"][i].numpy()
        G = o.shape[1]
        np.testing.assert_allclose(o, ref[row:row + len(t), :G], rtol=RTOL, atol=ATOL)
        assert np.all(ref[row:row + len(t), G:] == 0)     # pad_input zero fill
        row += len(t)


def test_shared_cre_stream_is_exact_dedup(golden):
    """The CRE stream is tissue-independent: evaluating it once per gene gives the
    reference's numbers."""
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, hp = _hps(meta)
    a = O.predict_step(batch, sd, cre_hp, gene_hp, hp, share_cre_stream=True)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(a["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(a["embeddings"][i], arrays[f"embeddings_{i}"], rtol=RTOL, atol=ATOL)


def test_mean_pool_before_the_last_down_projection_is_exact(golden):
    """The encoder's mean pool commutes with the last layer's linear_geglu_2 (seq2reg/model.py:263-267, seq2reg/modules.py:
    186-188): pooled first, the tokenizer embeddings equal the REFERENCE's own (fixture cre_tok / gene_tok) to fp32
    summation-order level, in pure fp32 and under the kernels' rounding points -- the re-ordering variantformer_amd executes
    (FlashTransformerLayer._pooled_down_projection) is algebra, not an approximation."""
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, _ = _hps(meta)
    with torch.no_grad():
        for i in range(len(meta["n_cres"])):
            for pfx, hp_, ids, mask, key in (("cre_tokenizer.", cre_hp, batch["cre_sequences"][i], batch["cre_attention_masks"][i], "cre_tok"),
                                             ("gene_tokenizer.", gene_hp, batch["gene_embeddings"][i], batch["gene_attention_masks"][i], "gene_tok")):
                early = O.seq2reg_embed(ids, mask, sd, pfx, hp_, O.Rounding(None), pool_before_down=True)
                np.testing.assert_allclose(early.numpy(), arrays[f"{key}_{i}"], rtol=RTOL, atol=ATOL)
                late = O.seq2reg_embed(ids, mask, sd, pfx, hp_, O.Rounding(None))
                assert float((early - late).abs().max()) < 2e-6 * float(late.abs().max())
                for mode in ("bf16", "fp16"):
                    e16 = O.seq2reg_embed(ids, mask, sd, pfx, hp_, O.Rounding(mode), pool_before_down=True)
                    l16 = O.seq2reg_embed(ids, mask, sd, pfx, hp_, O.Rounding(mode))
                    assert float((e16 - l16).abs().max()) < 2e-6 * float(l16.abs().max()), mode


def test_bf16_rounding_mode_is_close_to_fp32(golden):
    """The kernel-contract mode (bf16 operands, fp32 everything else) stays within bf16-level
    distance of the fp32 reference outputs; documents the precision of the shipped arithmetic."""
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, hp = _hps(meta)
    out = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding="bf16", share_cre_stream=True)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=3e-2, atol=3e-2)


def test_folded_layernorm_is_the_same_algebra(golden):
    """The LayerNorm -> Linear fold the bf16 HIP path uses (oracle.linear on a pending LayerNorm) is exact algebra: with
    no operand rounding it reproduces the reference outputs like the plain form; with bf16 operands it stays at
    bf16-level distance."""
    meta, arrays, sd, batch = golden
    cre_hp, gene_hp, hp = _hps(meta)
    for share in (False, True):
        out = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding=O.Rounding(None, fold_ln=True), share_cre_stream=share)
        for i in range(len(meta["n_cres"])):
            np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(out["embeddings"][i], arrays[f"embeddings_{i}"], rtol=RTOL, atol=ATOL)
    a = O.predict_step(batch, sd, cre_hp, gene_hp, hp, rounding=O.Rounding("bf16", fold_ln=True), share_cre_stream=True)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(a["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=3e-2, atol=3e-2)


def test_alibi_pe_precision_known_answers():
    misc = np.load(os.path.join(GOLDEN, "misc.npz"))
    for h in (2, 4, 8, 12, 32):
        np.testing.assert_allclose(np.array(O.alibi_slopes(h)), misc[f"alibi_{h}"], rtol=1e-6)
    # a-priori known answer: H=32 -> 2^(-(k+1)/4)
    np.testing.assert_allclose(O.alibi_slopes(32), [2 ** (-(k + 1) / 4) for k in range(32)], rtol=1e-12)
    np.testing.assert_array_equal(O.positional_encoding_1d(128, 40).numpy(), misc["pe_128_40"])
    np.testing.assert_array_equal(O.positional_encoding_1d(512, 200).numpy(), misc["pe_512_200"])
    with open(os.path.join(GOLDEN, "misc.json")) as f:
        prec = json.load(f)["precision2dtype"]
    for s, want in prec.items():
        if want.startswith("ERR"):
            try:
                O.precision2dtype(s)
                raise AssertionError("expected ValueError")
            except ValueError:
                pass
        else:
            assert str(O.precision2dtype(s)) == want


def test_pad_unpad_conventions():
    x = torch.arange(2 * 4 * 3, dtype=torch.float32).view(2, 4, 3)
    keep = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 0]], dtype=torch.bool)
    packed, idx, cu, mx, _ = O.unpad_input(x, keep)
    assert idx.tolist() == [0, 1, 4, 5, 6] and cu.tolist() == [0, 2, 5] and mx == 3 and cu.dtype == torch.int32
    back = O.pad_input(packed, idx, 2, 4)
    assert torch.equal(back[keep], x[keep]) and float(back[~keep].abs().sum()) == 0.0


import pytest
from tests.conftest import load_fixture


@pytest.mark.parametrize("name", ["small_opts_a", "small_opts_b", "small_opts_c", "small_opts_d"])
def test_oracle_matches_reference_with_non_shipped_options(name):
    """only_cross_attention + use_res + cross_alibi + start_token pooling (a); add_context_to_cres + max pooling (b);
    per-tissue small MLP heads (c); context-free CRE layers + shared linear head (d): outputs and intermediates of the
    reference's own classes with these options switched on."""
    meta, arrays, sd, batch = load_fixture(name)
    cre_hp, gene_hp, hp = _hps(meta)
    if name in ("small_opts_c", "small_opts_d"):      # these keep the tissue-independent CRE stream: dedup stays exact
        out = O.predict_step(batch, sd, cre_hp, gene_hp, hp, share_cre_stream=True)
        for i in range(len(meta["n_cres"])):
            np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(out["embeddings"][i], arrays[f"embeddings_{i}"], rtol=RTOL, atol=ATOL)
        out = O.predict_step(batch, sd, cre_hp, gene_hp, hp)
        for i in range(len(meta["n_cres"])):
            np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
        return
    assert not hp.shipped
    col = {}
    with torch.no_grad():
        pred, emb = O.forward(batch, sd, cre_hp, gene_hp, hp, collect=col)
    out = O.predict_step(batch, sd, cre_hp, gene_hp, hp)
    for i in range(len(meta["n_cres"])):
        np.testing.assert_allclose(out["pred_gene_exp"][i], arrays[f"pred_gene_exp_{i}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(out["embeddings"][i], arrays[f"embeddings_{i}"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(col["first_gene_layer_out"].numpy(), arrays["first_gene_layer_out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(col["first_cre_layer_out"].numpy(), arrays["first_cre_layer_out"], rtol=RTOL, atol=ATOL)
    with pytest.raises(AssertionError, match="de-duplicated"):
        O.predict_step(batch, sd, cre_hp, gene_hp, hp, share_cre_stream=True)


@pytest.mark.parametrize("name,share", [("vep_model", False), ("vep_model", True), ("vep_model_opts_a", False),
                                        ("vep_model_opts_b", False)])
def test_oracle_variant_prediction_matches_reference(name, share):
    """VEP token-position outputs: the oracle's variant_prediction against the reference's own
    variant_prediction run on CPU (tests/golden/vep_model*.*), with positions and with NaN positions; shipped options and
    two option sets (start token + cross-attention-only + use_res + cross_alibi; add_context_to_cres + max pooling)."""
    from tests.conftest import load_vep_model_fixture
    meta, arrays, sd, vb = load_vep_model_fixture(name)
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    ghp = O.Seq2GeneHP.from_kwargs(meta["seq2gene"])
    for tag, batch in (("pos", vb), ("nan", dict(vb, cre_token_position=torch.full((3, 1), float("nan")),
                                                 gene_token_position=torch.full((3, 1), float("nan"))))):
        out = O.variant_prediction(batch, sd, hp, hp, ghp, rounding=None, share_cre_stream=share)
        assert out["variant_type"] == meta["variant_type"]
        for k in ("pred_gene_exp", "embd", "gene_token_embedding", "cre_token_embedding"):
            for i in range(3):
                want = arrays[f"{tag}.{k}_{i}"]
                assert out[k][i].shape == want.shape == (len(meta["tissues"]), want.shape[1])
                np.testing.assert_allclose(out[k][i], want, rtol=2e-5, atol=2e-5)
    # the gathers really moved: het / hom rows differ from ref at the variant's window, and the NaN run is all zeros
    assert np.abs(arrays["pos.cre_token_embedding_1"] - arrays["pos.cre_token_embedding_0"]).max() > 1e-3
    assert np.abs(arrays["nan.gene_token_embedding_2"]).max() == 0.0


def test_precision_32_semantics_against_the_reference_fixture():
    """trainer.precision = "32": the reference casts its MHA modules (weights included, in place) to fp16 for their forward
    and keeps LayerNorms, GeGLU Linears and residuals in fp32 (seq2gene/modules/layers.py:98-126, utils/functions.py:28-30).
    variantformer_amd runs EVERY GEMM on fp16 operands in that mode.  The fixture small_sin_p32 is the
    reference's own run with precision "32" (tests/golden/make_golden.py); the fp16-rounding oracle -- the restatement of
    what the HIP path does -- lands within 1e-4 of it on the expression: the deviation is pinned, not only documented."""
    meta, arrays, sd, batch = load_fixture("small_sin_p32")
    assert meta["precision"] == "32"
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    out = O.predict_step(batch, sd, hp, hp, O.Seq2GeneHP.from_kwargs(meta["seq2gene"]), rounding="fp16", share_cre_stream=True)
    f32 = load_fixture("small_sin")            # same architecture, other seed: only used for the scale of "different"
    for i in range(len(meta["n_cres"])):
        want_p, want_e = arrays[f"pred_gene_exp_{i}"], arrays[f"embeddings_{i}"]
        assert float((np.abs(out["pred_gene_exp"][i] - want_p) / np.abs(want_p)).max()) < 1e-4       # measured 2.2e-5
        assert float(np.abs(out["embeddings"][i] - want_e).max() / np.abs(want_e).max()) < 1e-3      # measured 2.7e-4
    assert f32[0]["seed"] != meta["seed"]


def test_oracle_seq2reg_options_match_reference():
    """Tokenizer options outside the shipped configuration (seq_pool max / linear, use_context with and
    without expand_context, head dims 96 / 128): oracle vs the reference's own Seq2RegPredictor."""
    from tests.conftest import load_s2r_opts
    for name, (hp, want, sd, g) in load_s2r_opts().items():
        ohp = O.Seq2RegHP.from_hparams(hp)
        got = O.seq2reg_embed(g["cre_sequences"], g["cre_attention_masks"], sd, "", ohp, O.Rounding(None),
                              context=g["ref_cre_labels"])
        assert got.shape == want.shape
        np.testing.assert_allclose(got.numpy(), want, rtol=2e-5, atol=2e-5, err_msg=name)


def _data_kv_fixture():
    z = np.load(os.path.join(GOLDEN, "layer_data_kv.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "layer_data_kv.json")))
    src, ctx, mask = torch.from_numpy(z["src"]), torch.from_numpy(z["ctx"]), torch.from_numpy(z["mask"])
    keep = ~mask
    lens = keep.sum(dim=1)
    cu = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)]).to(torch.int32)
    return z, meta, src, ctx, mask, keep, cu


def data_kv_oracle(name, sd, xs, cs, cu, H, alibi, rnd):
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    if name == "cross":
        return O.cross_only_layer(xs, cs, cu, cu, sd, "", H, rnd, make_data_kv=True)
    return O.modulator_layer(xs, cs, cu, cu, sd, "", H, slopes, rnd, make_data_kv=True)


def test_make_data_kv_layers_match_the_reference_layer_classes():
    """make_data_kv=True (queries of the cross attention from the raw context, keys / values from the normalised stream;
    reference seq2gene/modules/layers.py:133-136,283-286, seq2reg/modules.py:97-100) is reachable only at layer level in the
    reference: the oracle's layer functions against the outputs of the reference's own three layer classes."""
    z, meta, src, ctx, mask, keep, cu = _data_kv_fixture()
    H = meta["nhead"]
    for name, info in meta["layers"].items():
        sd = {k[len(name) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + ".sd.")}
        with torch.no_grad():
            out = data_kv_oracle(name, sd, src[keep], ctx[keep], cu, H, info["use_alibi"], O.Rounding(None))
        np.testing.assert_allclose(out.numpy(), z[f"{name}.out"][keep.numpy()], rtol=RTOL, atol=ATOL)


def test_counted_key_attention_is_the_same_function_as_attention_over_the_repeated_keys():
    """The CRE layers' context rows are copies of <= 9 label embeddings: softmax over a gene's N context keys equals softmax over
    the distinct rows with log(count) added to the logit (oracle.attention_counted, the form vf_attn_counted_keys evaluates).
    In fp32 the two agree to round-off, with and without the base-2 pre-scaled q, for labels that are absent or occur once."""
    import math
    torch.manual_seed(5)
    H, dh, sq = 4, 48, 37
    labels = torch.tensor([3, 3, 0, 8, 3, 0, 3, 5, 3, 3, 0, 3])                   # labels 1, 2, 4, 6, 7 absent; 5 and 8 once
    tab_k, tab_v = torch.randn(9, H, dh), torch.randn(9, H, dh)
    q = torch.randn(sq, H, dh)
    uniq = sorted(set(labels.tolist()))
    counts = torch.tensor([int((labels == u).sum()) for u in uniq], dtype=torch.float32)
    for q_log2 in (False, True):
        qq = q * (math.log2(math.e) / math.sqrt(dh)) if q_log2 else q
        want = O.attention(qq, tab_k[labels], tab_v[labels], None, O.Rounding(None), q_log2=q_log2)
        got = O.attention_counted(qq, tab_k[uniq], tab_v[uniq], counts, q_log2)
        assert torch.allclose(got, want, rtol=2e-5, atol=2e-6), float((got - want).abs().max())
