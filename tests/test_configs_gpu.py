"""BASELINE.json configurations other than the bench workload, the flash-attn-compatible operator seam, and the
VEP token-position outputs, on a real MI355X through the C ABI.

  cfg 1  single gene, 128 kb window, 1 tissue   (N=40, C=178, T=1)          full depth vs oracle
  cfg 3  256 ragged genes of one donor, sharded by LPT over 8 ranks, batches of <= 8  full depth; oracle on a subset,
         shard / batch independence on all, reassembly in query order
  cfg 4  paired ref / alt 1 Mb windows                                                 full size: delta vs oracle delta
  seam   MHA.__call__ / FlashAttLayer.forward, packed and padded, self and cross       vs oracle.mha_self / mha_cross
  VEP    variant_prediction token-position outputs                                     vs the reference's own outputs

Error metric for new assertions: element-wise |a-b| / (|b| + rms(b)) (not max-norm), so that small elements count.
"""
import numpy as np
import pytest
import torch

from oracle import vf_oracle as O
from tests.conftest import load_vep_model_fixture
from tests.helpers import SEQ2REG_512, build_model, check_signal, erel, prel, seq2gene_kw, state_dict_cpu
from variantformer_amd.dist import gene_cost, shard_batch, shard_genes_lpt
from variantformer_amd.utils.synthetic import TISSUES_54, cfg3_gene_sizes, make_batch, make_vep_batch

pytestmark = pytest.mark.gpu

NORTH_STAR_RTOL = 1e-3     # expression output (BASELINE.json north star), element-wise
EMB_RTOL = 1e-2            # 1536-wide embedding rows after 49 bf16-operand layers, element-wise vs the same-rounding oracle
SEAM_RTOL = 1e-2           # one attention module: bf16 attention output (half an ulp = 2^-9) mixed by out_proj; the kernel
                           # rounds P against the running max, the oracle against the final max (same bound as
                           # tests/test_ops_gpu.py::test_attention_matches_oracle: rtol 2^-7)


@pytest.fixture(scope="module")
def full_model():
    import bench
    model, hp, kw = bench.build_model(torch.device("cuda", 0))
    return model, hp, kw, state_dict_cpu(model)


def _oracle(batch, sd, hp, kw, rounding="bf16"):
    import bench
    shp = O.Seq2RegHP.from_hparams(hp)
    torch.set_num_threads(min(16, bench.host_threads()))
    return O.predict_step(batch, sd, shp, shp, O.Seq2GeneHP.from_kwargs(kw), rounding=rounding, share_cre_stream=True)


# ------------------------------------------------------------------------------------------------------------------
def test_cfg1_single_gene_128kb_one_tissue(full_model):
    """BASELINE configs[0] geometry (SURVEY 8d cfg 1): N = 40 cCRE windows, C = 178 gene chunks, T = 1 tissue, the full
    25-layer / 6-layer architecture.  HIP vs the same-rounding oracle and vs the pure-fp32 oracle."""
    model, hp, kw, sd = full_model
    batch = make_batch(1281, [40], [178], [[33]], 200)
    out = model.predict_step(batch, 0)
    assert out["pred_gene_exp"][0].shape == (1, 1) and out["embeddings"][0].shape == (1, 1536)
    orc = _oracle(batch, sd, hp, kw)
    assert prel(out["pred_gene_exp"][0], orc["pred_gene_exp"][0]) < NORTH_STAR_RTOL
    assert erel(out["embeddings"][0], orc["embeddings"][0]) < EMB_RTOL
    f32 = _oracle(batch, sd, hp, kw, rounding=None)
    e_pred, e_emb = prel(out["pred_gene_exp"][0], f32["pred_gene_exp"][0]), erel(out["embeddings"][0], f32["embeddings"][0])
    print(f"[cfg1] HIP(bf16 operands) vs fp32 oracle, full depth: expression {e_pred:.2e}, embedding {e_emb:.2e} (element-wise)")
    # bf16 operands against pure fp32 arithmetic: the expression meets the north-star bar (1e-3 relative) directly; the
    # 1536-wide embedding row carries the bf16 noise of 49 layers (measured 7e-3 ... 1.1e-2)
    assert e_pred < NORTH_STAR_RTOL and e_emb < 2e-2
    # signal scale for a one-value output: the same gene over all 54 tissues (HIP), error of the one tissue against it
    spread = float(np.std(model.predict_step(dict(batch, tissue_context=[torch.tensor(TISSUES_54)]), 0)["pred_gene_exp"][0]))
    err = float(np.abs(out["pred_gene_exp"][0] - f32["pred_gene_exp"][0]).max())
    print(f"[signal] cfg1 vs pure fp32: |err| {err:.2e} = {err / spread:.2e} of the 54-tissue spread {spread:.3e}")
    assert err / spread < 5e-2


def test_cfg3_256_ragged_genes_lpt_shards(full_model):
    """BASELINE configs[2]: 256 genes of one donor (N ~ lognormal(600, 0.6) in [40, 2048], C ~ U{20..200}, 54 tissues),
    LPT-sharded over 8 ranks, each rank running batches of <= 8 genes; the expression matrix is reassembled in query
    order.  On one GPU the 8 shards run one after the other (the RCCL gather itself is covered by tests/test_dist_cpu.py
    and bench.py --workload cfg3).  Checks: (1) the three cheapest genes against the oracle, (2) every gene's result is
    independent of its shard / batch neighbours: a second pass with a different partition (round-robin, batches of 5)
    reproduces the matrix, (3) the matrix is finite, positive (Softplus) and in query order."""
    model, hp, kw, sd = full_model
    n_genes, T = 256, 54
    n, c = cfg3_gene_sizes(n_genes)
    costs = [gene_cost(int(a), int(b), T) for a, b in zip(n, c)]
    owned = shard_genes_lpt(costs, 8)

    def gene_batch(ids):
        return make_batch_by_gene(ids, n, c)

    def run(partition, bs):
        expr = np.full((n_genes, T), np.nan, np.float32)
        for shard in partition:
            for s in range(0, len(shard), bs):
                ids = shard[s:s + bs]
                out = model.predict_step(gene_batch(ids), 0)
                for j, g in enumerate(ids):
                    expr[g] = out["pred_gene_exp"][j][:, 0]
        return expr

    a = run(owned, 8)
    assert np.isfinite(a).all() and (a > 0).all()
    rr = [list(range(r, n_genes, 8)) for r in range(8)]
    b = run(rr, 5)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    cheapest = sorted(range(n_genes), key=lambda i: costs[i])[:3]
    orc = _oracle(gene_batch(cheapest), sd, hp, kw)
    for j, g in enumerate(cheapest):
        assert prel(a[g], orc["pred_gene_exp"][j][:, 0]) < NORTH_STAR_RTOL, (g, int(n[g]), int(c[g]))
    check_signal("cfg3, three genes x 54 tissues", [a[g] for g in cheapest], [orc["pred_gene_exp"][j][:, 0] for j in range(3)])
    loads = [sum(costs[i] for i in o) for o in owned]
    print(f"[cfg3] LPT imbalance max/mean = {max(loads) / (sum(loads) / 8):.4f}; genes per rank {[len(o) for o in owned]}")


def test_bench_step_of_32_headline_genes_equals_batches_of_8(full_model):
    """bench.py's step: 32 headline-size genes (1024 cCRE windows, 200 gene chunks, 54 tissues) in ONE pass -- 3.1 M seq2reg
    tokens, GEMM outputs of up to 4.7e9 elements (row offsets past 2^32 bytes) -- must give every gene the expression it
    gets in a batch of 8 (the reference DataLoader's batch size; the size every other full-size test runs at)."""
    from variantformer_amd.utils.synthetic import collate, make_gene
    model, hp, kw, sd = full_model
    genes = [make_gene(977 * 1000003 + g, 1024, 200, TISSUES_54, 200) for g in range(32)]
    with torch.no_grad():
        big = model.forward_prepared(model.prepare_batch(collate(genes)))[0].view(32, 54).float().cpu().numpy()
        parts = [model.forward_prepared(model.prepare_batch(collate(genes[s:s + 8])))[0].view(8, 54).float().cpu().numpy()
                 for s in range(0, 32, 8)]
    small = np.concatenate(parts)
    assert np.isfinite(big).all() and (big > 0).all()
    np.testing.assert_allclose(big, small, rtol=1e-5, atol=1e-6)
    assert len({tuple(r) for r in np.round(big, 5)}) == 32, "distinct genes must give distinct rows"
    torch.cuda.empty_cache()


def make_batch_by_gene(ids, n, c):
    """collate of the cfg-3 genes `ids`: gene g is always built from seed (20251205, g), whatever batch it lands in."""
    from variantformer_amd.utils.synthetic import collate, make_gene
    return collate([make_gene(20251205 * 1000003 + int(g), int(n[g]), int(c[g]), TISSUES_54, 200) for g in ids])


def test_cfg5_whole_genome_scan_fp16_ragged_genes(full_model):
    """BASELINE configs[4] workload ("whole-genome scan ... fp16 with fp32 accumulate"; SURVEY 8d cfg 5: 40 000 genes from
    the cfg-3 distribution): 32 genes of that draw (indices 256 ... 287, i.e. not the 256 of cfg 3), full depth,
    precision "16-mixed" -> fp16 operands.  (1) the three cheapest genes against oracle(rounding="fp16") and against pure
    fp32; (2) every gene is independent of its shard and batch: LPT shards of 8 ranks in batches of 8 vs round-robin
    shards in batches of 5; (3) finite, positive, query order."""
    model, hp, kw, sd = full_model
    T, first, count = 54, 256, 32
    n_all, c_all = cfg3_gene_sizes(40000)
    ids_all = list(range(first, first + count))
    n, c = {g: int(n_all[g]) for g in ids_all}, {g: int(c_all[g]) for g in ids_all}
    costs = [gene_cost(n[g], c[g], T) for g in ids_all]
    owned = [[ids_all[i] for i in o] for o in shard_genes_lpt(costs, 8)]

    def run(partition, bs):
        expr = {}
        for shard in partition:
            for s0 in range(0, len(shard), bs):
                ids = shard[s0:s0 + bs]
                out = model.predict_step(make_batch_by_gene(ids, n_all, c_all), 0)
                for j, g in enumerate(ids):
                    expr[g] = out["pred_gene_exp"][j][:, 0]
        return np.stack([expr[g] for g in ids_all])

    keep = model.precision
    model.precision = "16-mixed"
    try:
        assert model.operand_dtype() == torch.float16
        a = run(owned, 8)
        assert a.shape == (count, T) and np.isfinite(a).all() and (a > 0).all()
        b = run([ids_all[r::8] for r in range(8)], 5)
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
        cheapest = [ids_all[i] for i in sorted(range(count), key=lambda i: costs[i])[:3]]
        batch = make_batch_by_gene(cheapest, n_all, c_all)
        orc = _oracle(batch, sd, hp, kw, rounding="fp16")
        f32 = _oracle(batch, sd, hp, kw, rounding=None)
        for j, g in enumerate(cheapest):
            got = a[ids_all.index(g)]
            e16, e32 = prel(got, orc["pred_gene_exp"][j][:, 0]), prel(got, f32["pred_gene_exp"][j][:, 0])
            print(f"[cfg5] gene {g} (N={n[g]}, C={c[g]}): fp16-operand HIP vs oracle(fp16) {e16:.2e}, vs pure fp32 {e32:.2e}")
            assert e16 < NORTH_STAR_RTOL and e32 < NORTH_STAR_RTOL
        check_signal("cfg5 fp16, three genes x 54 tissues vs oracle(fp16)", [a[ids_all.index(g)] for g in cheapest],
                     [orc["pred_gene_exp"][j][:, 0] for j in range(3)])
    finally:
        model.precision = keep


def _variant_pair(seed, n_cre, n_chunk, tissues, n_var=5, whole_window=True):
    """ref gene + alt gene = the same gene with `n_var` cCRE windows and `n_var` gene chunks changed (BASELINE configs[3] /
    SURVEY 8d cfg 4: <= 5 + <= 5).  whole_window: every valid token of the window is re-drawn (an indel re-tokenises the
    window from the variant on); else 3 token ids per window (a substitution)."""
    ref = make_batch(seed, [n_cre], [n_chunk], [tissues], 200)
    pair = {k: (v + [t.clone() for t in v] if isinstance(v, list) else v.repeat(2, 1)) for k, v in ref.items()}
    rng = np.random.default_rng(7)
    for key, mkey, n in (("cre_sequences", "cre_attention_masks", n_cre), ("gene_embeddings", "gene_attention_masks", n_chunk)):
        for w in rng.choice(n, n_var, replace=False):
            valid = int((~pair[mkey][1][w, 0]).sum())
            if whole_window:
                pair[key][1][w, 0, :valid] = torch.from_numpy(rng.integers(4, 500, valid))
            else:
                pos = rng.choice(valid, 3, replace=False)
                pair[key][1][w, 0, pos] = torch.from_numpy(rng.integers(4, 18, 3))
    return ref, pair


def test_cfg4_paired_ref_alt_full_size(full_model):
    """BASELINE configs[3] (snp_indel_predictions geometry) at FULL SIZE: two 1 Mb genes (N = 1024, C = 200, T = 54) that
    differ in 5 cCRE windows and 5 gene chunks.  Structure: the ref half must not notice the alt half (pair vs alone), a
    tissue subset reproduces the full run, the VEP window de-duplication reproduces the plain evaluation bit for bit, and
    each half meets the north-star tolerance against the oracle.  The DELTA of this pair is reported, not asserted: with
    plain random weights the expression hardly depends on the sequence (utils.synthetic.calibrate_sequence_sensitivity has
    the measurement), |delta| ~ 1e-5 sits far below the bf16 rounding noise of ANY bf16 implementation -- resolving a
    variant effect is tested on a sequence-sensitive model in test_cfg4_variant_effect_is_resolved."""
    model, hp, kw, sd = full_model
    ref, pair = _variant_pair(40404, 1024, 200, TISSUES_54, whole_window=False)
    both = model.predict_step(pair, 0)
    alone = model.predict_step(ref, 0)
    np.testing.assert_allclose(both["pred_gene_exp"][0], alone["pred_gene_exp"][0], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        plain = model.forward_prepared(model.prepare_batch(pair, dedupe_windows=False))
        dd = model.prepare_batch(pair)                  # default: exact de-duplication on
        assert dd.cre_ids.shape[0] == 1024 + 5 and dd.gene_ids.shape[0] == 200 + 5
        dedup = model.forward_prepared(dd)
    assert torch.equal(plain[0], dedup[0]) and torch.equal(plain[1], dedup[1])
    # Oracle on both halves for a 6-tissue subset (a tissue's result does not depend on the other tissues requested --
    # asserted at this size by test_headline_size_gene_vs_oracle_and_properties -- and the oracle's cost is dominated by
    # the per-tissue gene stream): 2 x ~7.5 TFLOP on the host cores instead of 2 x 18.
    sub = [3, 11, 20, 31, 42, 53]
    pair6 = dict(pair, tissue_context=[t[sub] for t in pair["tissue_context"]])
    hip6 = model.predict_step(pair6, 0)
    for i in range(2):
        np.testing.assert_allclose(hip6["pred_gene_exp"][i], both["pred_gene_exp"][i][sub], rtol=1e-5, atol=1e-6)
    orc = _oracle(pair6, sd, hp, kw)
    for i in range(2):
        assert prel(hip6["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
    check_signal("cfg4 full size, ref + alt x 6 tissues", hip6["pred_gene_exp"], orc["pred_gene_exp"])
    d_hip = hip6["pred_gene_exp"][1] - hip6["pred_gene_exp"][0]
    d_orc = orc["pred_gene_exp"][1] - orc["pred_gene_exp"][0]
    noise = np.abs(d_hip - d_orc).max()
    print(f"[cfg4 full size, plain random weights] |delta_oracle| {np.abs(d_orc).max():.2e}, |delta_hip - delta_oracle| {noise:.2e}: "
          f"SNR {np.abs(d_orc).max() / noise:.2f} (reported only: the model is sequence-insensitive, see the docstring)")
    log2fc_h = np.log2(hip6["pred_gene_exp"][1] / hip6["pred_gene_exp"][0])
    log2fc_o = np.log2(orc["pred_gene_exp"][1] / orc["pred_gene_exp"][0])
    assert np.abs(log2fc_h - log2fc_o).max() < 1e-3          # the reference's own VEP tolerance (tests/test_vep.py atol 1e-3)


def test_cfg4_variant_effect_is_resolved(full_model):
    """The variant-effect delta must be RESOLVED, not merely bounded: on a sequence-sensitive model (the full 1.2 B
    architecture with cre_map / gene_map re-centred and rescaled, utils.synthetic.calibrate_sequence_sensitivity) and a
    locus in which 5 cCRE windows + 5 gene chunks are a visible share (N = 48, C = 20, T = 6), ref / alt differ by
    |log2FC| ~ 1e-2.  SNR = |delta_oracle| / |delta_hip - delta_oracle| against the same-rounding oracle AND against the
    pure-fp32 oracle must be >= 5 with bf16 operands and >= 20 with fp16 operands (measured 8 ... 10 and ~ 60: the bf16
    figure is the resolution of 8 mantissa bits, not of this implementation -- more gain in the calibration raises signal
    and rounding noise alike, scripts-free probe in DESIGN.md section 7); the log2FC must agree with fp32 to the
    reference's own 1e-3 (reference tests/test_vep.py:215-258) and have the right sign wherever |log2FC| > 2e-3."""
    import bench
    from variantformer_amd.utils.synthetic import calibrate_sequence_sensitivity, restore_maps
    model, hp, kw, _ = full_model
    saved = calibrate_sequence_sensitivity(model)
    keep = model.precision
    try:
        sd = state_dict_cpu(model)
        tissues = [TISSUES_54[i] for i in (3, 11, 20, 31, 42, 53)]
        _, pair = _variant_pair(50505, 48, 20, tissues)
        shp = O.Seq2RegHP.from_hparams(hp)
        ghp = O.Seq2GeneHP.from_kwargs(kw)
        torch.set_num_threads(min(16, bench.host_threads()))
        f32 = O.predict_step(pair, sd, shp, shp, ghp, rounding=None, share_cre_stream=True)["pred_gene_exp"]
        d32 = f32[1] - f32[0]
        l32 = np.log2(f32[1] / f32[0])
        print(f"[cfg4 sensitive model] fp32 oracle: max |delta| {np.abs(d32).max():.3e}, max |log2FC| {np.abs(l32).max():.3e}, "
              f"expression {f32[0].ravel()}")
        assert np.abs(l32).max() > 5e-3, "the calibrated model must show a variant effect worth resolving"
        for mode, prec in (("bf16", "bf16-mixed"), ("fp16", "16-mixed")):
            model.precision = prec
            hip = model.predict_step(pair, 0)["pred_gene_exp"]
            orc = O.predict_step(pair, sd, shp, shp, ghp, rounding=mode, share_cre_stream=True)["pred_gene_exp"]
            d_hip, d_orc = hip[1] - hip[0], orc[1] - orc[0]
            snr_same = np.abs(d_orc).max() / np.abs(d_hip - d_orc).max()
            snr_f32 = np.abs(d32).max() / np.abs(d_hip - d32).max()
            l_hip = np.log2(hip[1] / hip[0])
            print(f"[cfg4 sensitive model, {mode} operands] SNR vs same-rounding oracle {snr_same:.1f}, vs pure fp32 {snr_f32:.1f}; "
                  f"|log2FC_hip - log2FC_fp32| {np.abs(l_hip - l32).max():.2e}; halves vs same-rounding oracle "
                  f"{max(prel(hip[i], orc[i]) for i in range(2)):.2e}")
            # measured on MI355X (gpurun_out/r3d): bf16 7.9 vs the same-rounding oracle / 10.3 vs pure fp32 -- what 8
            # mantissa bits resolve of a |log2FC| ~ 9e-3 effect through 55 layers, for ANY bf16-operand implementation;
            # fp16 operands (3 more bits) must clear 20
            floor = 5.0 if mode == "bf16" else 20.0
            assert snr_same >= floor and snr_f32 >= floor, \
                f"{mode}: variant effect not resolved (SNR {snr_same:.1f} vs same-rounding oracle, {snr_f32:.1f} vs fp32)"
            # log2FC against fp32: the reference's own VEP tolerance (1e-3, tests/test_vep.py) holds with fp16 operands; with
            # bf16 operands this deliberately high-gain model sits at 0.9e-3 ... 1.2e-3 (its expression itself is only
            # good to 1.1e-3 against fp32, see `calibrate_sequence_sensitivity`), so the bf16 bound is 2e-3
            assert np.abs(l_hip - l32).max() < (2e-3 if mode == "bf16" else 1e-3)
            assert np.sign(l_hip[np.abs(l32) > 2e-3]).tolist() == np.sign(l32[np.abs(l32) > 2e-3]).tolist()   # direction of effect
    finally:
        model.precision = keep
        restore_maps(model, saved)


# ------------------------------------------------------------------------------------------------------------------
def _mha_sd(mha, prefix=""):
    return {prefix + k: v.detach().cpu().float() for k, v in mha.state_dict().items()}


@pytest.mark.parametrize("D,H,alibi", [(1536, 32, True), (512, 8, False), (128, 4, True)])
def test_mha_seam_self_packed_and_padded(D, H, alibi):
    """flash_attn.modules.mha.MHA call contract (reference seq2gene/modules/layers.py:437-439,465,482-487;
    seq2reg/modules.py:167): __call__(x, cu_seqlens=, max_seqlen=) on packed [tokens, D], and the padded [B, S, D]
    call without keyword arguments; through FlashAttLayer.forward as well (unpad_info and key-padding-mask forms)."""
    from variantformer_amd.seq2gene.modules.layers import MHA, FlashAttLayer
    from variantformer_amd.utils.synthetic import fill_state_dict
    rnd = O.Rounding("bf16")
    layer = FlashAttLayer(D, H, use_alibi=alibi, cross_attn=False)
    fill_state_dict(layer, 17)
    sd = _mha_sd(layer.MHA)
    layer = layer.cuda()
    mha: MHA = layer.MHA
    lens = [37, 1, 64, 130]
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(sum(lens), D, generator=g)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    want = O.mha_self(rnd.r(x), sd, "", H, cu, slopes, rnd)
    got = mha(x.cuda(), cu_seqlens=cu.cuda(), max_seqlen=max(lens))
    assert got.dtype == torch.float32 and got.shape == x.shape
    assert erel(got.cpu().numpy(), want.numpy()) < SEAM_RTOL
    got_l = layer(x.cuda(), unpad_info={"cu_seqlens": cu.cuda(), "max_seqlen": max(lens)})
    assert torch.equal(got_l, got)
    # bf16 in -> bf16 out, like the original module under autocast
    got_b = mha(x.cuda().bfloat16(), cu_seqlens=cu.cuda(), max_seqlen=max(lens))
    assert got_b.dtype == torch.bfloat16
    assert erel(got_b.float().cpu().numpy(), want.numpy()) < 1.5e-2
    # padded [B, S, D] without kwargs: every row is a key (no masking), seq2gene/modules/layers.py:487
    B, S = 3, 40
    xp = torch.randn(B, S, D, generator=g)
    cu_p = torch.arange(0, B + 1, dtype=torch.int32) * S
    want_p = O.mha_self(rnd.r(xp.reshape(B * S, D)), sd, "", H, cu_p, slopes, rnd).view(B, S, D)
    got_p = mha(xp.cuda())
    assert got_p.shape == (B, S, D)
    assert erel(got_p.cpu().numpy(), want_p.numpy()) < SEAM_RTOL
    # padded + key padding mask through FlashAttLayer (pack, run, scatter back, zeros on the pad rows)
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[0, 25:] = True
    mask[2, 1:] = True
    keep = ~mask
    cu_m = torch.tensor(np.concatenate([[0], np.cumsum(keep.sum(1).numpy())]), dtype=torch.int32)
    want_m = O.mha_self(rnd.r(xp[keep]), sd, "", H, cu_m, slopes, rnd)
    got_m = layer(xp.cuda(), src_key_padding_mask=mask.cuda())
    assert erel(got_m.cpu()[keep].numpy(), want_m.numpy()) < SEAM_RTOL
    assert float(got_m.cpu()[mask].abs().max()) == 0.0


@pytest.mark.parametrize("D,H", [(1536, 32), (128, 4)])
def test_mha_seam_cross_packed_and_padded(D, H):
    """Cross form: __call__(x, x_kv, cu_seqlens=, max_seqlen=, cu_seqlens_k=, max_seqlen_k=) and the padded call
    (reference seq2gene/modules/layers.py:437-439, 482-483)."""
    from variantformer_amd.seq2gene.modules.layers import FlashAttLayer
    from variantformer_amd.utils.synthetic import fill_state_dict
    rnd = O.Rounding("bf16")
    layer = FlashAttLayer(D, H, use_alibi=False, cross_attn=True)
    fill_state_dict(layer, 23)
    sd = _mha_sd(layer.MHA)
    layer = layer.cuda()
    q_lens, k_lens = [5, 201, 64], [300, 17, 1]
    cu_q = torch.tensor(np.concatenate([[0], np.cumsum(q_lens)]), dtype=torch.int32)
    cu_k = torch.tensor(np.concatenate([[0], np.cumsum(k_lens)]), dtype=torch.int32)
    g = torch.Generator().manual_seed(5)
    x, ctx = torch.randn(sum(q_lens), D, generator=g), torch.randn(sum(k_lens), D, generator=g)
    want = O.mha_cross(rnd.r(x), rnd.r(ctx), sd, "", H, cu_q, cu_k, rnd)
    got = layer.MHA(x.cuda(), ctx.cuda(), cu_seqlens=cu_q.cuda(), max_seqlen=max(q_lens), cu_seqlens_k=cu_k.cuda(),
                    max_seqlen_k=max(k_lens))
    assert erel(got.cpu().numpy(), want.numpy()) < SEAM_RTOL
    got_l = layer(x.cuda(), ctx.cuda(), unpad_info={"cu_seqlens": cu_q.cuda(), "max_seqlen": max(q_lens)},
                  context_unpad_info={"cu_seqlens": cu_k.cuda(), "max_seqlen": max(k_lens)})
    assert torch.equal(got_l, got)
    B, S, Sk = 2, 33, 70
    xp, cp = torch.randn(B, S, D, generator=g), torch.randn(B, Sk, D, generator=g)
    want_p = O.mha_cross(rnd.r(xp.reshape(-1, D)), rnd.r(cp.reshape(-1, D)), sd, "", H,
                         torch.arange(0, B + 1, dtype=torch.int32) * S, torch.arange(0, B + 1, dtype=torch.int32) * Sk, rnd)
    got_p = layer.MHA(xp.cuda(), cp.cuda())
    assert got_p.shape == (B, S, D)
    assert erel(got_p.cpu().numpy(), want_p.view(B, S, D).numpy()) < SEAM_RTOL
    with pytest.raises(AssertionError, match="context_key_padding_mask"):
        layer(xp.cuda(), cp.cuda(), src_key_padding_mask=torch.zeros(B, S, dtype=torch.bool).cuda())


def test_attention_rows_without_keys_are_zero():
    """A query sequence whose key sequence is empty gets zero rows (flash-attn's convention), not uninitialised memory."""
    from variantformer_amd import ops
    H, dh = 4, 48
    q = torch.randn(70, H * dh, device="cuda").bfloat16()
    kv = torch.randn(30, 2 * H * dh, device="cuda").bfloat16()
    cu_q = torch.tensor([0, 20, 70], dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0, 30, 30], dtype=torch.int32, device="cuda")
    out = torch.full((70, H * dh), float("nan"), device="cuda").bfloat16()
    ops.attn_varlen(q, kv[:, :H * dh], kv[:, H * dh:], cu_q, cu_k, 50, 30, H, dh, out=out)
    assert torch.isfinite(out[:20].float()).all() and float(out[:20].float().abs().max()) > 0
    assert float(out[20:].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------------------
def test_variant_prediction_vs_reference_golden():
    """pred_gene_exp, embd, gene_token_embedding and cre_token_embedding of the HIP path against
    the outputs of the reference's own variant_prediction (tests/golden/vep_model.*, fp32 CPU run) and against the
    same-rounding oracle; NaN positions give zero token embeddings."""
    meta, arrays, sd, vb = load_vep_model_fixture()
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    model.vep = True
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    ghp = O.Seq2GeneHP.from_kwargs(meta["seq2gene"])
    out = model.predict_step(vb, 0)
    orc = O.variant_prediction(vb, sd, hp, hp, ghp, rounding="bf16", share_cre_stream=True)
    assert out["variant_type"] == meta["variant_type"]
    for i in range(3):
        assert prel(out["pred_gene_exp"][i], arrays[f"pos.pred_gene_exp_{i}"]) < NORTH_STAR_RTOL
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        for k, ref_tol in (("embd", 1e-2), ("gene_token_embedding", 1e-2), ("cre_token_embedding", 1e-2)):
            got = out[k][i]
            assert got.shape == arrays[f"pos.{k}_{i}"].shape
            assert erel(got, arrays[f"pos.{k}_{i}"]) < ref_tol, (k, i)          # bf16 operands vs the fp32 reference run
            assert erel(got, orc[k][i]) < EMB_RTOL, (k, i)                         # same rounding points
    # the three genotypes really differ at the variant's windows
    assert np.abs(out["cre_token_embedding"][1] - out["cre_token_embedding"][0]).max() > 1e-3
    assert np.abs(out["gene_token_embedding"][2] - out["gene_token_embedding"][0]).max() > 1e-3
    nan_b = dict(vb, cre_token_position=torch.full((3, 1), float("nan")), gene_token_position=torch.full((3, 1), float("nan")))
    out_nan = model.predict_step(nan_b, 0)
    for i in range(3):
        assert float(np.abs(out_nan["gene_token_embedding"][i]).max()) == 0.0
        assert float(np.abs(out_nan["cre_token_embedding"][i]).max()) == 0.0
        np.testing.assert_array_equal(out_nan["pred_gene_exp"][i], out["pred_gene_exp"][i])


@pytest.mark.parametrize("name", ["vep_model_opts_a", "vep_model_opts_b"])
def test_variant_prediction_option_sets_vs_reference_golden(name):
    """variant_prediction on option sets the shipped configuration leaves off -- (a) one shared start token,
    cross-attention-only gene layers, gene residual, ALiBi on the cross attention; (b) tissue embedding added to the CRE
    tokens, max pooling (no token in front of the chunks: the +1 of reference :665-666 does not apply) -- all four outputs
    against the reference's own variant_prediction (tests/golden/vep_model_opts_*.*) and the same-rounding oracle."""
    meta, arrays, sd, vb = load_vep_model_fixture(name)
    model = build_model(meta["seq2reg"], meta["seq2gene"], sd).cuda()
    model.vep = True
    hp = O.Seq2RegHP.from_hparams(meta["seq2reg"])
    ghp = O.Seq2GeneHP.from_kwargs(meta["seq2gene"])
    out = model.predict_step(vb, 0)
    orc = O.variant_prediction(vb, sd, hp, hp, ghp, rounding="bf16", share_cre_stream=False)
    for i in range(3):
        assert prel(out["pred_gene_exp"][i], arrays[f"pos.pred_gene_exp_{i}"]) < NORTH_STAR_RTOL
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        for k in ("embd", "gene_token_embedding", "cre_token_embedding"):
            got = out[k][i]
            assert got.shape == arrays[f"pos.{k}_{i}"].shape
            # the bounds of tests/test_model_gpu.py::test_non_shipped_options_vs_reference_golden for these small fixtures
            # (max norm; max pooling keeps per-column extremes of bf16-noisy rows): bf16 operands vs the fp32 reference
            # run, then vs the oracle with the same rounding points
            mx = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
            assert mx(got, arrays[f"pos.{k}_{i}"]) < 2e-2, (k, i)
            assert mx(got, orc[k][i]) < (1e-2 if meta["seq2gene"]["gene_pooling"] == "max" else 5e-3), (k, i)
    assert np.abs(out["cre_token_embedding"][1] - out["cre_token_embedding"][0]).max() > 1e-3
    nan_b = dict(vb, cre_token_position=torch.full((3, 1), float("nan")), gene_token_position=torch.full((3, 1), float("nan")))
    out_nan = model.predict_step(nan_b, 0)
    for i in range(3):
        assert float(np.abs(out_nan["gene_token_embedding"][i]).max()) == 0.0
        assert float(np.abs(out_nan["cre_token_embedding"][i]).max()) == 0.0


def test_variant_prediction_production_width_vs_oracle():
    """Same outputs at production widths (D = 1536, H = 32, seq2reg d = 512), 3 modulator layers, 200-token windows."""
    kw = seq2gene_kw(layers=3)
    model = build_model(SEQ2REG_512, kw, seed=77)
    sd = state_dict_cpu(model)
    model = model.cuda()
    model.vep = True
    vb = make_vep_batch(515, 9, 6, TISSUES_54[:3], 200, cre_index=4, gene_index=(2, 2, 3))
    out = model.predict_step(vb, 0)
    hp = O.Seq2RegHP.from_hparams(SEQ2REG_512)
    orc = O.variant_prediction(vb, sd, hp, hp, O.Seq2GeneHP.from_kwargs(kw), rounding="bf16", share_cre_stream=True)
    for i in range(3):
        assert prel(out["pred_gene_exp"][i], orc["pred_gene_exp"][i]) < NORTH_STAR_RTOL
        for k in ("embd", "gene_token_embedding", "cre_token_embedding"):
            # the suite's element-wise bound against the same-rounding oracle (measured here: 2.9e-3 ... 5.2e-3, the two
            # runs sitting 4e-3 ... 9e-3 from pure fp32 arithmetic: what separates them are uncorrelated 16-bit flips)
            assert erel(out[k][i], orc[k][i]) < EMB_RTOL, (k, i)
