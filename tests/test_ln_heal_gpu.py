"""The folded LayerNorm never hands back degraded numbers: when the statistics kernels flag a batch (a residual-stream row
with |mean| beyond ops.LN_FOLD_RATIO_LIMIT standard deviations, or an element a scaled fp16 copy could not hold) the model
recomputes that batch with the separate LayerNorm on fp32 rows before predict_step / variant_prediction return -- bit for
bit what a VF_LN_FOLD=0 run returns (reference: plain nn.LayerNorm, seq2gene/modules/layers.py:75-77,99-163, no such
regime).  Also the range half of the flag on the op level (round-3 advice: a 2e6 outlier against the fp16 trunk copy)."""
import logging
import math

import numpy as np
import pytest
import torch

from tests.helpers import SEQ2REG_512, build_model, seq2gene_kw
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def heal_state(monkeypatch):
    """The self-healing bookkeeping is PER MODEL since round 6 (M.heal_state(model): a sticky switch when 2 of the model's last
    16 finished batches alerted); only the explicit process-wide switch (layers.ln_fold_disable) needs isolating.  These tests
    are about the DEFAULT contract (fold on, fp16 trunk copy): they pin their own environment whatever the ambient one."""
    monkeypatch.delenv("VF_LN_FOLD", raising=False)
    monkeypatch.delenv("VF_TRUNK16", raising=False)
    from variantformer_amd.seq2gene import model_combined_modulator as M
    from variantformer_amd.seq2gene.modules import layers as L
    saved = L._LN_FOLD_DISABLED
    L.ln_fold_reenable()
    yield M, L
    L._LN_FOLD_DISABLED = saved


def _forward_counter(model, monkeypatch):
    calls = {"n": 0}
    orig = model.forward_prepared

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(model, "forward_prepared", counted)
    return calls


def test_alerting_batch_is_recomputed_and_equals_the_unfolded_run(heal_state, monkeypatch, caplog):
    M, L = heal_state
    from variantformer_amd import ops
    tissues = [TISSUES_54[:3], [9]]
    kw = seq2gene_kw(layers=2)                       # production widths: D = 1536 / 32 heads, seq2reg 512 / 8 heads
    model = build_model(SEQ2REG_512, kw, seed=4242).cuda()
    batch = make_batch(99, [12, 5], [5, 3], tissues, 200)
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.ln_fold_alert(dev)                                             # clear whatever earlier tests left
    calls = _forward_counter(model, monkeypatch)

    # an ordinary batch: one forward, nothing flagged
    clean = model.predict_step(batch, 0)
    hs = M.heal_state(model)
    assert calls["n"] == 1 and hs.batches == 0

    # registry rows of the tissues in use get a mean of 20 standard deviations: the first gene layer's LayerNorm sees them
    with torch.no_grad():
        w = model.start_tkn.registry_tokens.weight
        for t in sorted({t for ts in tissues for t in ts}):
            w[t] += 20.0 * w[t].std()
    monkeypatch.setenv("VF_LN_FOLD", "0")
    calls["n"] = 0
    plain = model.predict_step(batch, 0)                               # the reference for "what must come back"
    assert calls["n"] == 1
    monkeypatch.delenv("VF_LN_FOLD")
    assert ops.ln_fold_alert(dev) == 0                                 # the unfolded path raises nothing

    calls["n"] = 0
    with caplog.at_level(logging.INFO, logger="variantformer_amd"):
        healed = model.predict_step(batch, 0)
    assert calls["n"] == 2, "the flagged batch must have been recomputed once"
    assert hs.batches == 1 and any("recomputed" in r.getMessage() for r in caplog.records)
    assert ops.ln_fold_alert(dev) == 0                                 # flag consumed
    for i in range(len(tissues)):
        assert np.array_equal(healed["pred_gene_exp"][i], plain["pred_gene_exp"][i])
        assert np.array_equal(healed["embeddings"][i], plain["embeddings"][i])
        assert not np.array_equal(healed["pred_gene_exp"][i], clean["pred_gene_exp"][i])     # the weights did change
    assert L.ln_fold_enabled(1536, 1024)                               # one alert: the fold is still on for the next batch

    # a second alerting batch switches the fold off FOR THIS MODEL: from then on ONE forward per batch, same bits
    calls["n"] = 0
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="variantformer_amd"):      # the switch is announced at WARNING: visible by default
        again = model.predict_step(batch, 1)
    assert calls["n"] == 2 and hs.batches == 2 and hs.off
    assert L.ln_fold_enabled(1536, 1024), "the switch is the model's own: nothing process-wide changed"
    assert any(r.levelno == logging.WARNING and "OFF" in r.getMessage() for r in caplog.records)
    st = model.ln_fold_state()
    assert st["switched_off_by_alerts"] and not st["switched_off_for_process"] and not st["enabled"] and st["batches_recomputed"] == 2
    # a second, healthy model in the same process keeps the fold (round-5 advice: the sticky switch was process-global)
    other = build_model(SEQ2REG_512, kw, seed=4243).cuda()
    seen = {}
    orig_fp = other._forward_prepared
    monkeypatch.setattr(other, "_forward_prepared", lambda *a, **k: (seen.update(fold=L.ln_fold_enabled(1536, 1024)), orig_fp(*a, **k))[1])
    other.predict_step(batch, 0)
    assert seen["fold"] is True and not other.ln_fold_state()["switched_off_by_alerts"]
    seen.clear()
    orig_m = model._forward_prepared
    monkeypatch.setattr(model, "_forward_prepared", lambda *a, **k: (seen.update(fold=L.ln_fold_enabled(1536, 1024)), orig_m(*a, **k))[1])
    calls["n"] = 0
    third = model.predict_step(batch, 2)
    assert calls["n"] == 1 and seen["fold"] is False
    for i in range(len(tissues)):
        assert np.array_equal(again["pred_gene_exp"][i], plain["pred_gene_exp"][i])
        assert np.array_equal(third["pred_gene_exp"][i], plain["pred_gene_exp"][i])
        assert np.array_equal(third["embeddings"][i], plain["embeddings"][i])


def test_pipelined_trainer_heals_the_right_batch(heal_state, monkeypatch):
    """Trainer.predict enqueues batch i, prepares batch i + 1 and only then finishes batch i: the flag read in
    predict_finish belongs to batch i (nothing of batch i + 1 has been launched), and only that batch is recomputed."""
    M, L = heal_state
    from variantformer_amd import ops
    from variantformer_amd.processors.trainer import Trainer
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=4242).cuda()
    good = make_batch(5, [6], [3], [[8, 9]], 200)
    bad = make_batch(6, [4], [2], [[30]], 200)
    # token 499 occurs in the middle batch only, and its embedding row carries a mean of ~28 standard deviations of the
    # encoder input row (embedding + sinusoidal table, std ~0.7): the seq2reg input statistics (width 512) flag that batch.
    # (An anomalous REGISTRY row would flag every batch: gene layer 0 projects the whole 63-row table once per batch.)
    for b in (good, bad):
        for v in b["cre_sequences"] + b["gene_embeddings"]:
            v[v == 499] = 498
    assert not bool(bad["cre_attention_masks"][0][0, 0, 5])
    bad["cre_sequences"][0][0, 0, 5] = 499
    with torch.no_grad():
        model.cre_tokenizer.token_embedding.weight[499] += 20.0
    ops.ln_fold_alert(torch.device("cuda", torch.cuda.current_device()))
    monkeypatch.setenv("VF_LN_FOLD", "0")
    want = Trainer().predict(model, [good, bad, good])
    monkeypatch.delenv("VF_LN_FOLD")
    monkeypatch.setattr(M, "LN_HEAL_STICKY_AFTER", 10 ** 9)
    calls = _forward_counter(model, monkeypatch)
    got = Trainer().predict(model, [good, bad, good])
    assert calls["n"] == 4 and M.heal_state(model).batches == 1          # three batches + one recomputation
    assert np.array_equal(got[1]["pred_gene_exp"][0], want[1]["pred_gene_exp"][0])
    assert np.array_equal(got[1]["embeddings"][0], want[1]["embeddings"][0])
    # the clean batches came from the folded path: same numbers as the unfolded run at 16-bit level, not bit for bit
    for i in (0, 2):
        np.testing.assert_allclose(got[i]["pred_gene_exp"][0], want[i]["pred_gene_exp"][0], rtol=1e-2)
        assert [r["batch_idx"] for r in got] == [0, 1, 2]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_range_bit_of_the_flag(dtype, monkeypatch):
    """An element of 2e6 cannot be held by fp16(x * 2^-4): the producer's statistics bound it (|mean| + sqrt(D var)) and
    raise bit 1 -- for the fp16 trunk copy of a bf16 model and for the scaled fp16 operand copy of an fp16 one; a bf16-only
    configuration (VF_TRUNK16=0) has no fp16 copy and stays silent; ordinary rows never raise it."""
    from variantformer_amd import ops
    monkeypatch.delenv("VF_TRUNK16", raising=False)                    # starts from the default (fp16 trunk copy) whatever the ambient switch
    monkeypatch.delenv("VF_LN_FOLD", raising=False)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    dev = torch.device("cuda", torch.cuda.current_device())
    M_, N, K = 300, 1536, 1024
    g = torch.Generator().manual_seed(3)
    a = (torch.rand((M_, K), generator=g) * 2 - 1).cuda().to(td)
    w = ((torch.rand((N, K), generator=g) * 2 - 1) / math.sqrt(K)).cuda().to(td)
    res = (torch.rand((M_, N), generator=g) * 2 - 1).cuda()
    with ops.compute_dtype(td):
        ops.ln_fold_alert(dev)
        s = ops.gemm_ln_producer(a, w, None, res)
        torch.cuda.synchronize()
        assert ops.ln_fold_alert(dev) == 0
        res2 = res.clone()
        res2[17, 100] = 2.0e6
        s = ops.gemm_ln_producer(a, w, None, res2)
        bits = ops.ln_fold_alert(dev)
        assert bits & 2, bits
        assert ops.ln_fold_alert(dev) == 0                             # reading resets
        # the stream-input forms raise it too
        ops.ln_stream(res2)
        assert ops.ln_fold_alert(dev) & 2
        if dtype == "bf16":
            t = ops.trunk16_of(res2)
            assert ops.ln_fold_alert(dev) & 2 and torch.isinf(t[17, 100])
            monkeypatch.setenv("VF_TRUNK16", "0")                      # no fp16 copy anywhere: nothing to overflow
            ops.gemm_ln_producer(a, w, None, res2)
            assert ops.ln_fold_alert(dev) & 2 == 0


def test_deeper_pipelining_heals_the_right_batch(heal_state, monkeypatch):
    """Round-4 advice: a driver that launches batch i + 1 BEFORE finishing batch i.  The alert travels with the batch (cut out
    of the stream's flag behind the batch's last kernel, carried in the opaque handle), so finish(i) neither recomputes for
    batch i + 1's rows nor lets batch i + 1 return unhealed numbers."""
    M, L = heal_state
    from variantformer_amd import ops
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=4242).cuda().eval()
    good = make_batch(5, [6], [3], [[8, 9]], 200)
    bad = make_batch(6, [4], [2], [[30]], 200)
    for b in (good, bad):
        for v in b["cre_sequences"] + b["gene_embeddings"]:
            v[v == 499] = 498
    bad["cre_sequences"][0][0, 0, 5] = 499
    with torch.no_grad():
        model.cre_tokenizer.token_embedding.weight[499] += 20.0
    monkeypatch.setenv("VF_LN_FOLD", "0")
    want = [model.predict_step(b, i) for i, b in enumerate([good, bad, good])]
    monkeypatch.delenv("VF_LN_FOLD")
    monkeypatch.setattr(M, "LN_HEAL_STICKY_AFTER", 10 ** 9)
    # a stale flag of a direct forward() call (no batch owns it) must not count against the next batch either
    with torch.no_grad():
        model.forward_prepared(model.prepare_batch(bad))
    calls = _forward_counter(model, monkeypatch)
    hs = M.heal_state(model)
    hs.batches = hs.finished = 0
    handles = [model.predict_launch(model.prepare_batch(b)) for b in (good, bad, good)]      # three launches in flight
    got = [model.predict_finish(h, i) for i, h in enumerate(handles)]
    assert calls["n"] == 4 and hs.batches == 1 and hs.finished == 3
    assert list(hs.recent)[-3:] == [False, True, False]
    assert np.array_equal(got[1]["pred_gene_exp"][0], want[1]["pred_gene_exp"][0])
    assert np.array_equal(got[1]["embeddings"][0], want[1]["embeddings"][0])
    for i in (0, 2):
        np.testing.assert_allclose(got[i]["pred_gene_exp"][0], want[i]["pred_gene_exp"][0], rtol=1e-2)
    # and out of order: finishing the flagged batch LAST still heals it, not its neighbours
    calls["n"] = 0
    handles = [model.predict_launch(model.prepare_batch(b)) for b in (bad, good)]
    g1 = model.predict_finish(handles[1], 1)
    g0 = model.predict_finish(handles[0], 0)
    assert calls["n"] == 3
    assert np.array_equal(g0["pred_gene_exp"][0], want[1]["pred_gene_exp"][0])
    np.testing.assert_allclose(g1["pred_gene_exp"][0], want[0]["pred_gene_exp"][0], rtol=1e-2)


def test_an_unused_registry_row_does_not_flag_the_batch(heal_state, monkeypatch):
    """Gene layer 0 projects the whole 63-row registry table once per batch; only the rows of the tissues the batch asks for
    may raise the alert (round-4 advice: one anomalous row of an unused tissue would otherwise flag every batch and switch the
    fold off for the process)."""
    M, L = heal_state
    kw = seq2gene_kw(layers=2)
    model = build_model(SEQ2REG_512, kw, seed=4242).cuda().eval()
    batch = make_batch(7, [6], [3], [[8, 9]], 200)
    with torch.no_grad():
        w = model.start_tkn.registry_tokens.weight
        w[40] += 30.0 * w[40].std()                                      # tissue 40: not in the batch
    calls = _forward_counter(model, monkeypatch)
    model.predict_step(batch, 0)
    assert calls["n"] == 1 and M.heal_state(model).batches == 0
    other = make_batch(7, [6], [3], [[8, 40]], 200)                      # now it is
    model.predict_step(other, 1)
    assert calls["n"] == 3 and M.heal_state(model).batches == 1


def test_forced_off_is_thread_local_and_flags_are_per_stream(heal_state):
    """A recomputation in one thread must not flip the path of a forward in another; kernels raise the flag of the stream
    they run on."""
    import threading
    M, L = heal_state
    from variantformer_amd import ops
    seen = {}
    with L.ln_fold_forced_off():
        assert not L.ln_fold_enabled(1536, 1024)
        t = threading.Thread(target=lambda: seen.update(other=L.ln_fold_enabled(1536, 1024)))
        t.start(); t.join()
    assert seen["other"] is True and L.ln_fold_enabled(1536, 1024)
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.ln_fold_alert(dev)
    x = torch.randn(64, 1536, device=dev)
    x[3] += 100.0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.ln_stream(x)
        side.synchronize()
        assert ops.ln_fold_alert(dev, reset=False) & 1                   # raised on the side stream ...
    torch.cuda.synchronize()
    assert ops.ln_fold_alert(dev) == 0                                   # ... and invisible to the default stream
    with torch.cuda.stream(side):
        assert ops.ln_fold_alert(dev) & 1 and ops.ln_fold_alert(dev) == 0
