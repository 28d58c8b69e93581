"""Kernels of two HIP streams in flight together give the results they give alone.

Round 6 found (scripts/probes/concurrency_probe3-5.py, pk_hazard_probe.hip, profiles/r06_d_*): a packed-fp32 VALU instruction
whose op_sel takes the high dword of src1 (v_pk_fma_f32 ... op_sel:[0,1,0]: the LayerNorm-consumer epilogue of gemm_mfma_kernel as
hipcc's SLP vectoriser emits it) computes with a wrong src1 in lanes 48..63 while a wave of another kernel on the same SIMD issues
MFMAs -- the output element becomes mean-term + bias, a few units of 16 rows x 1 column per launch beside the cross attention.  The library is therefore built WITHOUT packed-fp32 instructions
(csrc/build.py NO_PACKED_FP32, tests/test_build_flags.py); these tests keep two streams busy with the pairs that went wrong and compare bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cu(lens, dev):
    return torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)


@pytest.fixture(scope="module")
def setup():
    from variantformer_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)

    def rnd(*shape, scale=1.0):
        return (torch.rand(shape, device=dev, generator=g) * 2 - 1) * scale
    G, K, H, dh = 4, 1536, 32, 48
    D = H * dh
    ql, kl = [54 * 201] * G, [1024] * G
    q, kv = rnd(sum(ql), D, scale=0.35).bfloat16(), rnd(sum(kl), 2 * D).bfloat16()
    cq, ck = _cu(ql, dev), _cu(kl, dev)
    s = ops.ln_stream(rnd(1024 * G, K))
    w, b, c = (rnd(4608, K) / K ** 0.5).bfloat16(), rnd(4608), rnd(4608)
    w320, b320, c320 = (rnd(320, K) / K ** 0.5).bfloat16(), rnd(320), rnd(320)
    a = rnd(1024 * G, K).bfloat16()
    victims = {
        "consumer16": lambda: ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16),
        "consumer_geglu": lambda: ops.gemm_ln_consumer(s, w, b, c, ops.EPI_GEGLU_BF16),
        "consumer_logits": lambda: ops.gemm_ln_consumer(s, w320, b320, c320, ops.EPI_F32),
        "plain128": lambda: ops.gemm(a, w, b, ops.EPI_BF16, variant=1),
    }

    def cross(x32):
        import os
        os.environ["VF_ATTN_X32"] = "1" if x32 else "0"
        try:
            return ops.attn_varlen(q, kv[:, :D], kv[:, D:], cq, ck, max(ql), max(kl), H, dh, None, q_log2=True)
        finally:
            os.environ.pop("VF_ATTN_X32", None)
    return ops, victims, cross


@pytest.mark.parametrize("victim", ["consumer16", "consumer_geglu", "consumer_logits", "plain128"])
@pytest.mark.parametrize("x32", [True, False], ids=["beside_attn_32x32x16", "beside_attn_16x16x32"])
def test_gemm_beside_attention_on_another_stream_is_bit_identical(setup, victim, x32):
    ops, victims, cross = setup
    f = victims[victim]
    with torch.no_grad():
        ref = f().clone()
        coref = cross(x32).clone()
        torch.cuda.synchronize()
        main, side = torch.cuda.current_stream(), torch.cuda.Stream()
        for _ in range(4):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                outs = [f() for _ in range(3)]
            couts = [cross(x32) for _ in range(2)]
            main.wait_stream(side)
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref), f"{victim}: {int((o != ref).sum())} elements differ beside the attention kernel"
            for o in couts:
                assert torch.equal(o, coref)


def test_two_models_in_two_threads_on_their_own_streams():
    """Two model instances driven from two threads, each on its own HIP stream (and, by default, its own CRE side stream): every
    result equals the one the model gives alone, bit for bit, while the other model's kernels share the GPU."""
    import threading
    from tests.helpers import SEQ2REG_512, build_model, seq2gene_kw
    from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
    models = [build_model(SEQ2REG_512, seq2gene_kw(layers=6), seed=21 + i).cuda() for i in range(2)]
    batches = [make_batch(31 + i, [280 + 40 * i, 64], [120, 30 + 5 * i], [TISSUES_54[:9], TISSUES_54[3:8]], 200) for i in range(2)]
    alone = [m.predict_step(b, 0) for m, b in zip(models, batches)]          # (also the cache-building first forwards)
    alone = [m.predict_step(b, 0) for m, b in zip(models, batches)]
    errors, results = [], [[], []]
    gate = threading.Barrier(2)

    def work(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                gate.wait()
                for _ in range(6):
                    results[i].append(models[i].predict_step(batches[i], 0))
            stream.synchronize()
        except Exception as e:                                    # noqa: BLE001 (re-raised in the main thread)
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert len(results[i]) == 6
        for out in results[i]:
            for g in range(2):
                np.testing.assert_array_equal(out["pred_gene_exp"][g], alone[i]["pred_gene_exp"][g])
                np.testing.assert_array_equal(out["embeddings"][g], alone[i]["embeddings"][g])
