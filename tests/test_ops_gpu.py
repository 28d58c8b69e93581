"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle's arithmetic
(oracle/vf_oracle.py, bf16-operand mode) on seeded inputs.  Tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from variantformer_amd import ops as _ops
    from variantformer_amd import _lib
    _lib.load()      # must be the in-tree HIP library; raises if missing
    return _ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ---------------------------------------------------------------------------------------------
# GEMM: all epilogues, MFMA path (K % 64 == 0) and generic path, ragged M / N tails
# ---------------------------------------------------------------------------------------------
GEMM_SHAPES = [
    (1, 128, 64), (7, 8, 64), (128, 128, 128), (300, 192, 192), (257, 576, 192), (1000, 1536, 512),
    (130, 2048, 128), (64, 96, 96), (33, 288, 96), (513, 4608, 1536), (200, 1536, 1024), (77, 40, 72),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("epi", ["bf16", "f32", "res", "gelu_f32", "gelu_bf16"])
def test_gemm_epilogues(ops, M, N, K, epi):
    a = _bf(_rand((M, K), 1))
    w = _bf(_rand((N, K), 2, 1.0 / math.sqrt(K)))
    b = _rand((N,), 3, 0.5)
    res = _rand((M, N), 4)
    ref = a @ w.t() + b
    code = {"bf16": ops.EPI_BF16, "f32": ops.EPI_F32, "res": ops.EPI_RES_F32, "gelu_f32": ops.EPI_GELU_F32,
            "gelu_bf16": ops.EPI_GELU_BF16}[epi]
    if epi == "res":
        ref = ref + res
    if epi.startswith("gelu"):
        ref = F.gelu(ref)
    out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), code,
                   residual=res.cuda() if epi == "res" else None)
    torch.cuda.synchronize()
    got = out.float().cpu()
    if epi in ("bf16", "gelu_bf16"):
        # bf16 store: half an ulp of bf16 (2^-9 relative) on top of fp32 accumulation-order noise
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2 ** -8, atol=2e-3)
    else:
        # same bf16 operands, fp32 accumulate: only summation order differs
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,F2,K", [(5, 64, 64), (300, 2048, 192), (129, 2048, 512), (64, 64, 96), (1000, 2048, 1536)])
def test_gemm_geglu(ops, M, F2, K):
    a = _bf(_rand((M, K), 11))
    w = _bf(_rand((F2, K), 12, 1.0 / math.sqrt(K)))
    b = _rand((F2,), 13, 0.5)
    h = a @ w.t() + b
    x, gate = h.chunk(2, dim=-1)
    ref = x * F.gelu(gate)
    wp, bp = ops.pack_geglu_rows(w.cuda().bfloat16(), b.cuda())
    out = ops.gemm(a.cuda().bfloat16(), wp, bp, ops.EPI_GEGLU_BF16)
    torch.cuda.synchronize()
    assert out.shape == (M, F2 // 2)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=2 ** -8, atol=2e-3)


def test_gemm_linearity_full_size(ops):
    """Size-independent property at production shape: A @ W^T is linear in A (bf16-exact inputs)."""
    M, N, K = 8192, 4608, 1536
    a1 = torch.randint(-4, 5, (M, K), generator=torch.Generator().manual_seed(5)).float()
    a2 = torch.randint(-4, 5, (M, K), generator=torch.Generator().manual_seed(6)).float()
    w = torch.randint(-3, 4, (N, K), generator=torch.Generator().manual_seed(7)).float()
    wc = w.cuda().bfloat16()
    o1 = ops.gemm(a1.cuda().bfloat16(), wc, None, ops.EPI_F32)
    o2 = ops.gemm(a2.cuda().bfloat16(), wc, None, ops.EPI_F32)
    o12 = ops.gemm((a1 + a2).cuda().bfloat16(), wc, None, ops.EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(o1 + o2, o12)            # small integers: every partial sum is exact in fp32
    ref_rows = (a1[:64] @ w.t())
    assert torch.equal(o1[:64].cpu(), ref_rows)


def test_gemm_rejects_bad_shapes(ops):
    from variantformer_amd._lib import VFError
    a = torch.zeros((4, 20), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((8, 20), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(VFError):
        ops.gemm(a, w, None, ops.EPI_F32)           # K % 8 != 0
    with pytest.raises(VFError):
        ops.gemm(torch.zeros((4, 64), dtype=torch.bfloat16), torch.zeros((8, 64), dtype=torch.bfloat16), None, ops.EPI_F32)  # CPU tensors


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, cu_q, cu_k, H, dh, slopes):
    rnd = O.Rounding("bf16")
    out = torch.zeros(q.shape[0], H * dh)
    for b in range(len(cu_q) - 1):
        a, e = int(cu_q[b]), int(cu_q[b + 1])
        ka, ke = int(cu_k[b]), int(cu_k[b + 1])
        if e > a and ke > ka:
            out[a:e] = O.attention(q[a:e].view(-1, H, dh), k[ka:ke].view(-1, H, dh), v[ka:ke].view(-1, H, dh),
                                   slopes, rnd).reshape(e - a, H * dh)
    return out


ATTN_CASES = [
    # (dh, H, q lens, k lens (None = self), alibi)
    (64, 2, [5, 40, 1, 33], None, False),
    (64, 8, [200, 70, 125, 64, 65, 128, 129], None, True),
    (48, 4, [201, 201, 17], None, True),
    (48, 4, [300, 1], None, True),                       # > 256 -> 2 query groups per wave
    (48, 32, [603], [1024], False),                      # gene -> CRE cross attention, shared K/V
    (48, 4, [10, 50, 7], [9, 300, 64], False),
    (32, 2, [31, 64, 100], None, True),
    (64, 2, [1000], None, True),
    (48, 2, [130, 257], [1, 63], False),
    (48, 4, [201, 9, 64, 65], [100, 256, 1, 129], False),    # short-sequence kernel, cross attention, ragged
    (64, 8, [70, 125, 99, 1, 128], None, False),             # short-sequence kernel, 2 query groups per wave
    (48, 32, [201] * 6, None, True),                         # gene stream shape
    # >= 2048 blocks of 256 queries -> 4 query groups per wave (batched gene -> CRE cross attention), ragged
    (48, 32, [1300, 257, 600, 1024, 999, 256, 255, 1, 770, 512, 1100, 300], [300, 64, 100, 1, 129, 200, 65, 77, 256, 31, 128, 90], False),
    # head dims a tokenizer checkpoint may carry (d = 768 / 8 heads, d = 1024 / 8 heads): 256-byte K rows, 3 / 4 k-steps
    (96, 4, [70, 200, 33, 1], None, False),
    (96, 2, [300, 64], [77, 500], False),
    (96, 8, [125, 99], None, True),
    (128, 2, [130, 64, 1, 200], None, True),
    (128, 4, [257], [1000], False),
]


@pytest.mark.parametrize("dh,H,ql,kl,alibi", ATTN_CASES)
def test_attention_matches_oracle(ops, dh, H, ql, kl, alibi):
    self_attn = kl is None
    kl = ql if self_attn else kl
    D = H * dh
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    if self_attn:
        qkv = _bf(_rand((tq, 3 * D), 21, 2.0))
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dev = qkv.cuda().bfloat16()
        dq, dk, dv = dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:]
    else:
        q = _bf(_rand((tq, D), 22, 2.0))
        kv = _bf(_rand((tk, 2 * D), 23, 2.0))
        k, v = kv[:, :D], kv[:, D:]
        dq = q.cuda().bfloat16()
        dkv = kv.cuda().bfloat16()
        dk, dv = dkv[:, :D], dkv[:, D:]
    ref = _attn_ref(q, k, v, cu_q, cu_k, H, dh, slopes)
    out = ops.attn_varlen(dq, dk, dv, cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh,
                          slopes.cuda() if alibi else None)
    torch.cuda.synchronize()
    got = out.float().cpu()
    # bf16 output (2^-9) + bf16 P rounding at a different running max than the oracle's final max
    np.testing.assert_allclose(got.numpy(), _bf(ref).numpy(), rtol=2 ** -7, atol=6e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("alibi,cross", [(True, False), (False, False), (False, True)])
def test_attention_short_kernel_ragged_batch(ops, dtype, alibi, cross):
    """The one-block-per-(sequence, head) kernel (dh <= 48, 128 < max_q <= 256) on a ragged batch of 70 sequences: lengths
    1, 16-row and 64-row boundaries (every size of the 16 / 32 / 64-key tail tile), 256, an empty key sequence and an empty
    query sequence; both operand types, with ALiBi, without, and as a cross attention.  Deterministic across repeats."""
    dh, H = 48, 8
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    rng = np.random.default_rng(5)
    ql = [256, 201, 129, 1, 16, 17, 64, 65, 128, 192, 193, 255, 0, 80, 81, 96, 97, 208, 209, 224, 225] + list(rng.integers(1, 257, 49))
    kl = ql
    if cross:
        kl = [200, 256, 1, 63, 0, 64, 65, 129, 7, 255, 128, 31, 40, 16, 17, 32, 33, 48, 49, 80, 81] + list(rng.integers(1, 257, 49))
    D = H * dh
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    rd = (lambda t: t.to(tdt).float())
    if not cross:
        qkv = rd(_rand((tq, 3 * D), 31, 2.0))
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dev = qkv.cuda().to(tdt)
        dq, dk, dv = dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:]
    else:
        q = rd(_rand((tq, D), 32, 2.0))
        kv = rd(_rand((tk, 2 * D), 33, 2.0))
        k, v = kv[:, :D], kv[:, D:]
        dq = q.cuda().to(tdt)
        dkv = kv.cuda().to(tdt)
        dk, dv = dkv[:, :D], dkv[:, D:]
    args = (dq, dk, dv, cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh, slopes.cuda() if alibi else None)
    outs = [ops.attn_varlen(*args).clone() for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))
    rnd = O.Rounding(dtype)
    got = outs[0].float().cpu()
    for b in list(range(21)) + [30, 50]:
        a, e, ka, ke = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if e == a:
            continue
        if ke == ka:
            assert float(got[a:e].abs().max()) == 0.0
            continue
        ref = O.attention(q[a:e].view(-1, H, dh), k[ka:ke].view(-1, H, dh), v[ka:ke].view(-1, H, dh), slopes, rnd)
        np.testing.assert_allclose(got[a:e].numpy(), rd(ref.reshape(e - a, D)).numpy(), rtol=2 ** -7, atol=6e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("alibi", [True, False])
def test_attention_short_kernel_prescaled_q(ops, dtype, alibi):
    """VF_ATTN_Q_LOG2 on the one-block-per-(sequence, head) kernel (the gene stream's self attention): q carries the base-2
    softmax scale, the running maximum is rounded up to an integer (attn_tile mode 2).  Ragged batch (1 ... 256 tokens,
    every tail-tile size) in which one sequence has logits of several hundred, one has a single extreme query among
    ordinary ones, and one has every logit below -300; every element of those and of a sample of ordinary sequences
    against the oracle evaluated on the same pre-scaled, rounded q."""
    dh, H = 48, 8
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    rd = rnd.r
    c = math.log2(math.e) / math.sqrt(dh)
    rng = np.random.default_rng(6)
    ql = [256, 201, 129, 1, 16, 17, 64, 65, 200, 193, 255, 80] + list(rng.integers(1, 257, 20))
    D = H * dh
    tq = sum(ql)
    cu = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    x = _rand((tq, 3 * D), 51, 2.0)
    q, k, v = x[:, :D].clone(), rd(x[:, D:2 * D]), rd(x[:, 2 * D:])
    a1, e1 = int(cu[1]), int(cu[2])
    q[a1:e1] *= 14.0                                   # sequence 1: logits of several hundred, both signs
    q[int(cu[8]) + 77] *= 16.0                         # sequence 8: one extreme query
    a2, e2 = int(cu[2]), int(cu[3])
    base = _rand((1, D), 52, 2.0)
    k[a2:e2] = rd(base.repeat(e2 - a2, 1) * (1.0 + 0.01 * torch.arange(e2 - a2)[:, None]))
    q[a2:e2] = -7.0 * base                             # sequence 2: every logit << -126
    qs = rd(q * c)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    dev = torch.cat([qs, k, v], dim=1).cuda().to(tdt)
    out = ops.attn_varlen(dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:], cu.cuda(), None, max(ql), max(ql), H, dh,
                          slopes.cuda() if alibi else None, q_log2=True)
    again = ops.attn_varlen(dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:], cu.cuda(), None, max(ql), max(ql), H, dh,
                            slopes.cuda() if alibi else None, q_log2=True)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), again.view(torch.int16))
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = dict(rtol=2 ** -7, atol=6e-3) if dtype == "bf16" else dict(rtol=2 ** -9, atol=2e-3)
    for b in list(range(12)) + [15, 25]:
        a, e = int(cu[b]), int(cu[b + 1])
        ref = O.attention(qs[a:e].view(-1, H, dh), k[a:e].view(-1, H, dh), v[a:e].view(-1, H, dh), slopes, rnd, q_log2=True)
        np.testing.assert_allclose(got[a:e].numpy(), rd(ref.reshape(e - a, D)).numpy(), **tol)


@pytest.mark.parametrize("alibi", [True, False])
def test_attention_prescaled_q_is_kernel_independent(ops, alibi):
    """A query's result must not depend on which kernel its batch geometry selects (a gene must not change with the genes
    it is batched with): with VF_ATTN_Q_LOG2 the one-block-per-sequence kernel and the tiled kernels run the same
    arithmetic per query (attn_tile mode 2: one fused multiply-add for the ALiBi bias, integer running maxima), so they
    produce the same bits.  Two sequences alone (short kernel) and next to a 300-token one (tiled kernel, 2 query groups
    per wave); one sequence alone (tiled kernel, 64-query blocks) and next to a 201-token one (short kernel)."""
    dh, H = 48, 8
    D = H * dh
    c = math.log2(math.e) / math.sqrt(dh)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32).cuda() if alibi else None
    ql = [201, 150, 300]
    x = _bf(_rand((sum(ql), 3 * D), 61, 2.0))
    x[:, :D] = _bf(x[:, :D] * c)
    dev = x.cuda().bfloat16()

    def run(lens):
        n = sum(lens)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
        return ops.attn_varlen(dev[:n, :D], dev[:n, D:2 * D], dev[:n, 2 * D:], cu, None, max(lens), max(lens), H, dh, slopes,
                               q_log2=True)
    short = run(ql[:2])                                   # max_q = 201: attn_short_kernel, no maximum
    tiled = run(ql)                                       # max_q = 300: attn_fwd_kernel
    torch.cuda.synchronize()
    assert torch.equal(short.view(torch.int16), tiled[:351].view(torch.int16))
    small = run([100])                                    # max_q = 100: attn_fwd_kernel, one query group per wave
    alone = run([100, 201])[:100]
    assert torch.equal(small.view(torch.int16), alone.view(torch.int16))


@pytest.mark.parametrize("q_log2", [True, False])
@pytest.mark.parametrize("dh,alibi", [(64, False), (32, False), (32, True)])
def test_attention_one_block_per_window_dh64(ops, q_log2, dh, alibi):
    """seq2reg windows (dh = 64, <= 128 tokens) in a batch of >= 1024 (window, head) items take one block per item with the
    whole K / V in LDS (attn_short2_kernel<64, .., NPASS = 1>, three resident blocks per CU) instead of the tiled kernel's two
    64-query blocks.  130 ragged windows of 1 ... 128 tokens: every element against the oracle, and bit-identical to the
    tiled kernel (the same windows in a batch too small for the new path): the arithmetic per query is the same.
    dh = 32 (round 6: a tokenizer geometry the real checkpoint might have takes the same kernel), with and without ALiBi."""
    H = 8
    D = H * dh
    slopes_l = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None      # (CPU copy for the oracle)
    slopes = slopes_l.cuda() if alibi else None
    rng = np.random.default_rng(9)
    ql = [128, 127, 1, 16, 17, 32, 33, 64, 65, 96, 97, 112, 113, 80] + list(rng.integers(1, 129, 116))
    assert len(ql) * H >= 1024
    c = math.log2(math.e) / math.sqrt(dh) if q_log2 else 1.0
    x = _bf(_rand((sum(ql), 3 * D), 71, 2.0))
    x[:, :D] = _bf(x[:, :D] * c)
    dev = x.cuda().bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    out = ops.attn_varlen(dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:], cu.cuda(), None, max(ql), max(ql), H, dh, slopes, q_log2=q_log2)
    assert ops.last_kernel("attn") == "attn_short2_kernel<1 pass>"
    n_small = 100                                              # 100 windows x 8 heads < 1024 items: the tiled kernel
    t_small = int(cu[n_small])
    small = ops.attn_varlen(dev[:t_small, :D], dev[:t_small, D:2 * D], dev[:t_small, 2 * D:], cu[:n_small + 1].cuda(), None,
                            max(ql[:n_small]), max(ql[:n_small]), H, dh, slopes, q_log2=q_log2)
    assert ops.last_kernel("attn").startswith("attn_fwd_kernel")
    torch.cuda.synchronize()
    if q_log2 or not alibi:      # (without the pre-scaled q the ALiBi bias meets the logit in a different expression per kernel)
        assert torch.equal(out[:t_small].view(torch.int16), small.view(torch.int16))
    rnd = O.Rounding("bf16")
    got = out.float().cpu()
    for b in list(range(14)) + [40, 90, 129]:
        a, e = int(cu[b]), int(cu[b + 1])
        ref = O.attention(x[a:e, :D].view(-1, H, dh), x[a:e, D:2 * D].view(-1, H, dh), x[a:e, 2 * D:].view(-1, H, dh), slopes_l, rnd,
                          q_log2=q_log2)
        np.testing.assert_allclose(got[a:e].numpy(), _bf(ref.reshape(e - a, D)).numpy(), rtol=2 ** -7, atol=6e-3)


@pytest.mark.parametrize("alibi", [False, True])
def test_attention_dh96_windows_as_one_128_query_block(ops, alibi):
    """dh = 96 (a tokenizer geometry the real checkpoint might have, e.g. d = 768 with 8 heads): windows of 65 ... 128 tokens in
    a batch of >= 1024 (window, head) items run as ONE 128-query block of the tiled kernel (round 6; two query groups per wave)
    instead of two 64-query blocks that each stage all keys.  Every element of a few windows against the oracle, and
    bit-identical to the 64-query form (the same windows in a batch too small for the new path)."""
    H, dh = 8, 96
    D = H * dh
    slopes_l = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    slopes = slopes_l.cuda() if alibi else None
    rng = np.random.default_rng(19)
    ql = [128, 127, 65, 66, 96, 97, 112, 113, 80, 1, 17, 64] + list(rng.integers(1, 129, 120))
    assert len(ql) * H >= 1024
    c = math.log2(math.e) / math.sqrt(dh)
    x = _bf(_rand((sum(ql), 3 * D), 73, 2.0))
    x[:, :D] = _bf(x[:, :D] * c)
    dev = x.cuda().bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    out = ops.attn_varlen(dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:], cu.cuda(), None, max(ql), max(ql), H, dh, slopes, q_log2=True)
    assert ops.last_kernel("attn") == "attn_fwd_kernel<128-query blocks>", ops.last_kernel("attn")
    n_small = 100
    t_small = int(cu[n_small])
    small = ops.attn_varlen(dev[:t_small, :D], dev[:t_small, D:2 * D], dev[:t_small, 2 * D:], cu[:n_small + 1].cuda(), None,
                            max(ql[:n_small]), max(ql[:n_small]), H, dh, slopes, q_log2=True)
    assert ops.last_kernel("attn") == "attn_fwd_kernel<64-query blocks>", ops.last_kernel("attn")
    torch.cuda.synchronize()
    assert torch.equal(out[:t_small].view(torch.int16), small.view(torch.int16))
    rnd = O.Rounding("bf16")
    got = out.float().cpu()
    for b in list(range(12)) + [40, 131]:
        a, e = int(cu[b]), int(cu[b + 1])
        ref = O.attention(x[a:e, :D].view(-1, H, dh), x[a:e, D:2 * D].view(-1, H, dh), x[a:e, 2 * D:].view(-1, H, dh), slopes_l, rnd,
                          q_log2=True)
        np.testing.assert_allclose(got[a:e].numpy(), _bf(ref.reshape(e - a, D)).numpy(), rtol=2 ** -7, atol=6e-3)


@pytest.mark.parametrize("q_log2", [True, False])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_attention_one_block_per_chunk_dh64_two_passes(ops, q_log2, dtype):
    """seq2reg's gene chunks (dh = 64, 129 ... 256 tokens; 200 in the shipped configuration) in a batch of >= 1024 (chunk, head)
    items take one block per item in the two-pass form of attn_short2_kernel (62 KB image, two resident blocks per CU) instead
    of four 64-query blocks of the tiled kernel.  Ragged chunks incl. the boundaries 129 / 192 / 193 / 200 / 256: bit-identical
    to the tiled kernel (the same chunks in a batch too small for the new path), sampled chunks against the oracle."""
    dh, H = 64, 8
    D = H * dh
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    rng = np.random.default_rng(19)
    ql = [200, 129, 256, 192, 193, 255, 130, 200, 224, 160] + list(rng.integers(129, 257, 120))
    assert len(ql) * H >= 1024
    c = math.log2(math.e) / math.sqrt(dh) if q_log2 else 1.0
    x = rnd.r(_rand((sum(ql), 3 * D), 73, 2.0))
    x[:, :D] = rnd.r(x[:, :D] * c)
    dev = x.cuda().to(td)
    cu = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    out = ops.attn_varlen(dev[:, :D], dev[:, D:2 * D], dev[:, 2 * D:], cu.cuda(), None, max(ql), max(ql), H, dh, q_log2=q_log2)
    n_small = 100                                              # 100 chunks x 8 heads < 1024 items: the tiled kernel
    t_small = int(cu[n_small])
    small = ops.attn_varlen(dev[:t_small, :D], dev[:t_small, D:2 * D], dev[:t_small, 2 * D:], cu[:n_small + 1].cuda(), None,
                            max(ql[:n_small]), max(ql[:n_small]), H, dh, q_log2=q_log2)
    torch.cuda.synchronize()
    assert torch.equal(out[:t_small].view(torch.int16), small.view(torch.int16))
    got = out.float().cpu()
    tol = dict(rtol=2 ** -7, atol=6e-3) if dtype == "bf16" else dict(rtol=2 ** -9, atol=2e-3)
    for b in list(range(10)) + [60, 129]:
        a, e = int(cu[b]), int(cu[b + 1])
        ref = O.attention(x[a:e, :D].view(-1, H, dh), x[a:e, D:2 * D].view(-1, H, dh), x[a:e, 2 * D:].view(-1, H, dh), None, rnd,
                          q_log2=q_log2)
        np.testing.assert_allclose(got[a:e].numpy(), rnd.r(ref.reshape(e - a, D)).numpy(), **tol)


def test_attention_online_softmax_rescale_branch(ops):
    """Force the running max to jump at a late key tile (guide rule 26): one key far larger than the rest."""
    dh, H, n = 64, 1, 200
    g = torch.Generator().manual_seed(3)
    q = _bf(torch.randn(n, dh, generator=g))
    k = _bf(torch.randn(n, dh, generator=g) * 0.1)
    v = _bf(torch.randn(n, dh, generator=g))
    k[150] = q[7] * 4.0                      # spike in the third key tile for query 7 (and large for others)
    cu = torch.tensor([0, n], dtype=torch.int32)
    ref = _attn_ref(q, k, v, cu, cu, H, dh, None)
    out = ops.attn_varlen(q.cuda().bfloat16(), k.cuda().bfloat16(), v.cuda().bfloat16(), cu.cuda(), cu.cuda(), n, n, H, dh)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), _bf(ref).numpy(), rtol=2 ** -7, atol=6e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_attention_long_stream_32x32_kernel(ops, dtype):
    """The long-stream dh = 48 kernel on 32x32x16 MFMAs (attn_x32_kernel; taken when n_seq * H * ceil(max_q / 256) >= 2048:
    the batched gene -> CRE cross attention).  Ragged query blocks (not multiples of 32 / 64 / 256), key streams of 1, 31,
    63, 64, 65 and 1024 keys (tail masking inside the first 32-key block, at a block edge, one past it), a sequence
    without keys (zero rows), a sequence of one query; every output element against the oracle.  A spike key in a LATE
    tile forces the online-softmax rescale branch with non-trivial accumulators (guide rule 26)."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    dh, H = 48, 32
    D = H * dh
    ql = [5000, 33, 2049, 777, 1, 3000, 1024, 900, 256, 257, 4000, 31]
    kl = [1024, 1, 63, 64, 200, 65, 0, 31, 129, 1000, 300, 2]
    assert len(ql) * H * ((max(ql) + 255) // 256) >= 2048
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32)
    q = rnd.r(_rand((tq, D), 24, 2.0))
    kv = rnd.r(_rand((tk, 2 * D), 25, 2.0))
    k, v = kv[:, :D].clone(), kv[:, D:].clone()
    # spike: key 900 of sequence 0 is aligned with its query 4321 in every head -> the running maximum of that query
    # (and of many others) jumps in key tile 14 of 16
    k[900] = rnd.r(q[4321] * 1.5)
    ref = torch.zeros(tq, D)
    for b in range(len(ql)):
        a, e, ka, ke = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if ke > ka:
            ref[a:e] = O.attention(q[a:e].view(-1, H, dh), k[ka:ke].view(-1, H, dh), v[ka:ke].view(-1, H, dh), None,
                                   rnd).reshape(e - a, D)
    dkv = torch.cat([k, v], dim=1).cuda().to(td)
    out = torch.full((tq, D), float("nan"), device="cuda").to(td)
    ops.attn_varlen(q.cuda().to(td), dkv[:, :D], dkv[:, D:], cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh, out=out)
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    z0, z1 = int(cu_q[6]), int(cu_q[7])
    assert float(got[z0:z1].abs().max()) == 0.0                       # the sequence without keys
    tol = dict(rtol=2 ** -7, atol=6e-3) if dtype == "bf16" else dict(rtol=2 ** -9, atol=2e-3)
    np.testing.assert_allclose(got.numpy(), rnd.r(ref).numpy(), **tol)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("big_batch", [True, False])
def test_attention_prescaled_q_without_running_maximum(ops, dtype, big_batch):
    """VF_ATTN_Q_LOG2 (q projected with weights pre-multiplied by softmax_scale * log2 e): the long-stream dh = 48 kernel
    takes p = exp2(q . k) with NO running maximum and recomputes, in the same launch, every block whose denominator left
    [2^-100, 2^100].  Ordinary sequences (fast path), a sequence whose logits reach +-400 (overflow -> recomputed), one
    whose logits are all below -300 (every key underflows -> recomputed), a mixed block (one extreme query among ordinary
    ones), ragged tails, an empty key sequence; 64 queries per wave (>= 2048 blocks) and 32 per wave; both operand types.
    Every element against the oracle evaluated on the same (pre-scaled, rounded) q."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    dh, H = 48, 32
    D = H * dh
    c = math.log2(math.e) / math.sqrt(dh)
    ql = [3000, 33, 2049, 700, 1, 2500, 600, 300, 256] if big_batch else [300, 33, 129, 1]
    kl = [1024, 70, 63, 64, 200, 65, 0, 31, 1000] if big_batch else [260, 1, 0, 64]
    if big_batch:
        assert len(ql) * H * ((max(ql) + 255) // 256) >= 2048
    else:
        assert len(ql) * H * ((max(ql) + 255) // 256) < 2048
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32)
    q = _rand((tq, D), 44, 2.0)
    kv = rnd.r(_rand((tk, 2 * D), 45, 2.0))
    k, v = kv[:, :D].clone(), kv[:, D:].clone()
    s1 = int(cu_q[1])
    q[s1:s1 + 33] *= 12.0                         # sequence 1: logits up to several hundred in either direction (overflow)
    if big_batch:
        # sequence 3: every key is the same direction and every query points the other way -> all logits << -126
        k3a, k3e = int(cu_k[3]), int(cu_k[4])
        base = _rand((1, D), 46, 2.0)
        k[k3a:k3e] = rnd.r(base.repeat(k3e - k3a, 1) * (1.0 + 0.01 * torch.arange(k3e - k3a)[:, None]))
        q3a, q3e = int(cu_q[3]), int(cu_q[4])
        q[q3a:q3e] = -6.0 * base
        q[100] *= 15.0                            # one extreme query inside an ordinary block of sequence 0
        # sequence 7: every logit around -12 -- harmless for bf16 P, but an fp16 P would be subnormal (4 significant bits)
        k7a, k7e = int(cu_k[7]), int(cu_k[8])
        base7 = _rand((1, D), 49, 1.0)
        k[k7a:k7e] = rnd.r(base7.repeat(k7e - k7a, 1) * (1.0 + 0.02 * torch.arange(k7e - k7a)[:, None]))
        q7a, q7e = int(cu_q[7]), int(cu_q[8])
        per_head = (base7.view(H, dh) ** 2).sum(dim=1)                      # q . k per head for q = base7
        q[q7a:q7e] = (base7.view(H, dh) * (-12.0 / (per_head * c))[:, None]).reshape(1, D).repeat(q7e - q7a, 1)
        # sequence 8: a THOUSAND keys with logits of -15 ... -19: their exponentials add up to ~2^-5.5, past a test on the
        # denominator alone, although every fp16 P would be subnormal; the mean-p test (l > len_k * 2^-11) recomputes the rows
        k8a, k8e = int(cu_k[8]), int(cu_k[9])
        base8 = _rand((1, D), 50, 1.0)
        k[k8a:k8e] = rnd.r(base8.repeat(k8e - k8a, 1) * (1.0 + 0.00027 * torch.arange(k8e - k8a)[:, None]))
        q8a, q8e = int(cu_q[8]), int(cu_q[9])
        per_head8 = (base8.view(H, dh) ** 2).sum(dim=1)
        q[q8a:q8e] = (base8.view(H, dh) * (-15.0 / (per_head8 * c))[:, None]).reshape(1, D).repeat(q8e - q8a, 1)
    qs = rnd.r(q * c)                             # what the pre-scaled Wq projection hands over (one rounding)
    ref = torch.zeros(tq, D)
    for b in range(len(ql)):
        a, e, ka, ke = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if ke > ka:
            ref[a:e] = O.attention(qs[a:e].view(-1, H, dh), k[ka:ke].view(-1, H, dh), v[ka:ke].view(-1, H, dh), None,
                                   rnd, q_log2=True).reshape(e - a, D)
    dkv = torch.cat([k, v], dim=1).cuda().to(td)
    out = torch.full((tq, D), float("nan"), device="cuda").to(td)
    ops.attn_varlen(qs.cuda().to(td), dkv[:, :D], dkv[:, D:], cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh, out=out,
                    q_log2=True)
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    tol = dict(rtol=2 ** -7, atol=6e-3) if dtype == "bf16" else dict(rtol=2 ** -9, atol=2e-3)
    np.testing.assert_allclose(got.numpy(), rnd.r(ref).numpy(), **tol)
    if big_batch:
        # The first four sequences alone are a batch for the 32-queries-per-wave form of the kernel (128-query blocks instead of
        # 256): a query's arithmetic is the same, so its bits are -- except inside a block that was RECOMPUTED with the running
        # maximum in one geometry and not in the other (the unit of recomputation is the block): exp2(s - m) and exp2(s) * 2^-m
        # differ in the last bit of P when s - m rounds (measured: 7 of 512 rows by one bf16 ulp).  Sequence 2 and the rows of
        # sequence 0 past its first 256 hold no extreme query: bit-identical; everything else within one ulp of the output.
        assert 4 * H * ((max(ql[:4]) + 255) // 256) < 2048
        nq4, nk4 = int(cu_q[4]), int(cu_k[4])
        sub = torch.full((nq4, D), float("nan"), device="cuda").to(td)
        ops.attn_varlen(qs[:nq4].cuda().to(td), dkv[:nk4, :D], dkv[:nk4, D:], cu_q[:5].cuda(), cu_k[:5].cuda(), max(ql[:4]),
                        max(kl[:4]), H, dh, out=sub, q_log2=True)
        torch.cuda.synchronize()
        s2a, s2e = int(cu_q[2]), int(cu_q[3])
        assert torch.equal(sub[256:ql[0]].view(torch.int16), out[256:ql[0]].view(torch.int16))
        assert torch.equal(sub[s2a:s2e].view(torch.int16), out[s2a:s2e].view(torch.int16))
        # (one ulp of one P moves an output by <= 2^-8 (bf16) / 2^-11 (fp16) of |v| ~ 2: absolute, the outputs cancel towards zero)
        np.testing.assert_allclose(sub.float().cpu().numpy(), out[:nq4].float().cpu().numpy(), rtol=2 ** -7 if dtype == "bf16" else 2 ** -10,
                                   atol=4e-3 if dtype == "bf16" else 5e-4)
    # the flag on the other kernels (here: dh = 64 cross attention, 16x16x32 tiles) only moves the scale
    dh2, H2 = 64, 4
    q2 = _rand((150, H2 * dh2), 47, 2.0)
    kv2 = rnd.r(_rand((90, 2 * H2 * dh2), 48, 2.0))
    c2 = math.log2(math.e) / math.sqrt(dh2)
    cu2q, cu2k = torch.tensor([0, 150], dtype=torch.int32), torch.tensor([0, 90], dtype=torch.int32)
    d2 = kv2.cuda().to(td)
    plain = ops.attn_varlen(rnd.r(q2).cuda().to(td), d2[:, :H2 * dh2], d2[:, H2 * dh2:], cu2q.cuda(), cu2k.cuda(), 150, 90, H2, dh2)
    pre = ops.attn_varlen(rnd.r(q2 * c2).cuda().to(td), d2[:, :H2 * dh2], d2[:, H2 * dh2:], cu2q.cuda(), cu2k.cuda(), 150, 90, H2,
                          dh2, q_log2=True)
    np.testing.assert_allclose(pre.float().cpu().numpy(), plain.float().cpu().numpy(), rtol=2 ** -5, atol=2e-2)


def test_attention_uniform_values_property(ops):
    """Size-independent property at full size: with V constant along keys the output equals that
    constant row exactly up to bf16 rounding of P (softmax weights sum to one)."""
    dh, H, nq, nk = 48, 32, 54 * 201, 1024
    D = H * dh
    q = _bf(_rand((nq, D), 31, 3.0)).cuda().bfloat16()
    k = _bf(_rand((nk, D), 32, 3.0)).cuda().bfloat16()
    row = _bf(_rand((1, D), 33, 2.0))
    v = row.repeat(nk, 1).cuda().bfloat16()
    cu_q = torch.tensor([0, nq], dtype=torch.int32).cuda()
    cu_k = torch.tensor([0, nk], dtype=torch.int32).cuda()
    out = ops.attn_varlen(q, k, v, cu_q, cu_k, nq, nk, H, dh)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), row.repeat(nq, 1).numpy(), rtol=2 ** -6, atol=1e-2)
    assert torch.isfinite(out.float()).all()


# ---------------------------------------------------------------------------------------------
# streaming kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,D", [(1, 96), (7, 128), (1000, 512), (333, 1536), (5, 2048), (3, 4096)])
@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm(ops, rows, D, gelu):
    x = _rand((rows, D), 41, 3.0) + 0.5
    g, b = 1 + 0.1 * _rand((D,), 42), 0.1 * _rand((D,), 43)
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    o32 = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), torch.float32, gelu)
    o16 = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), torch.bfloat16, gelu)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o32.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)    # fp32 throughout
    assert torch.equal(o16.cpu(), o32.cpu().bfloat16())                                  # same value, RNE to bf16


def test_embed_pack_and_cu_seqlens_bit_exact(ops):
    W, L, d, V = 37, 200, 128, 500
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, V, (W, L), generator=g)
    pad = torch.rand((W, L), generator=g) < 0.3           # arbitrary (non-suffix) masks, incl. edge windows
    pad[0] = True
    pad[0, 5] = False
    pad[1] = False
    pad[2] = True                                          # empty window
    table = _rand((V, d), 8)
    pos = _rand((L, d), 9)
    cu = ops.mask_to_cu_seqlens(pad.cuda())
    lens = (~pad).sum(1)
    ref_cu = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(torch.int32)
    assert torch.equal(cu.cpu(), ref_cu)
    n = int(ref_cu[-1])
    out = ops.embed_pack(ids.cuda(), pad.cuda(), cu, table.cuda(), pos.cuda(), n)
    out_nopos = ops.embed_pack(ids.cuda(), pad.cuda(), cu, table.cuda(), None, n)
    torch.cuda.synchronize()
    x = table[ids] + pos[None]
    keep = ~pad
    assert torch.equal(out.cpu(), x[keep])                 # one fp32 add: bit exact
    assert torch.equal(out_nopos.cpu(), table[ids][keep])


@pytest.mark.parametrize("d,dtype", [(512, "bf16"), (128, "bf16"), (1536, "bf16"), (512, "fp16")])
def test_embed_stream_equals_embed_pack_then_stream_passes(ops, d, dtype):
    """vf_embed_stream (embedding + positional table -> 16-bit operand copy, fp16 trunk copy, row statistics, no fp32 rows)
    against the three kernels it replaces -- vf_embed_pack, vf_row_stats_cast2, the fp16 trunk cast: bit-identical copies and
    statistics (same lane / column mapping, same reduction order); ragged, arbitrary (non-suffix) masks, an empty window."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    W, L, V = 41, 200, 500
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(0, V, (W, L), generator=g)
    pad = torch.rand((W, L), generator=g) < 0.4
    pad[0] = True
    pad[0, 7] = False
    pad[1] = False
    pad[2] = True
    table = _rand((V, d), 18, 2.0)
    pos = _rand((L, d), 19)
    with ops.compute_dtype(td):
        cu = ops.mask_to_cu_seqlens(pad.cuda())
        n = int(cu[-1])
        for p in (pos.cuda(), None):
            x = ops.embed_pack(ids.cuda(), pad.cuda(), cu, table.cuda(), p, n)
            want = ops.ln_stream(x)
            t_want = ops.trunk16_of(x)
            got = ops.embed_stream(ids.cuda(), pad.cuda(), cu, table.cuda(), p, n, need_x=True, need_t16=True)
            lean = ops.embed_stream(ids.cuda(), pad.cuda(), cu, table.cuda(), p, n, need_x=False, need_t16=False)
            torch.cuda.synchronize()
            assert got.scale == want.scale and torch.equal(got.x, x)
            assert got.x16.dtype == td and torch.equal(got.x16, want.x16) and torch.equal(got.stats, want.stats)
            assert torch.equal(got.t16, t_want)
            assert lean.x is None and lean.t16 is None and torch.equal(lean.x16, want.x16) and torch.equal(lean.stats, want.stats)


def test_mask_to_cu_seqlens_many_windows(ops):
    W, L = 5000, 40
    pad = torch.rand((W, L), generator=torch.Generator().manual_seed(1)) < 0.5
    cu = ops.mask_to_cu_seqlens(pad.cuda()).cpu()
    ref = torch.cat([torch.zeros(1, dtype=torch.int64), (~pad).sum(1).cumsum(0)]).to(torch.int32)
    assert torch.equal(cu, ref)


def test_segment_mean(ops):
    lens = [3, 1, 200, 0, 77]
    d = 512
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    x = _rand((sum(lens), d), 51, 2.0)
    o32 = ops.segment_mean(x.cuda(), cu.cuda(), torch.float32).cpu()
    o16 = ops.segment_mean(x.cuda(), cu.cuda(), torch.bfloat16).cpu()
    for w, n in enumerate(lens):
        a = int(cu[w])
        if n == 0:
            assert torch.isnan(o32[w]).all()               # reference: 0/0 -> NaN (seq2reg/model.py:264-267)
        else:
            # sequential fp32 sum in token order, then * (1/n)
            np.testing.assert_allclose(o32[w].numpy(), x[a:a + n].sum(0).numpy() / n, rtol=1e-5, atol=1e-6)
    ok = ~torch.isnan(o32)
    assert torch.equal(o16[ok], o32.bfloat16()[ok])


def test_gathers_bit_exact(ops):
    a, b = _rand((50, 192), 61), _rand((9, 192), 62)
    idx = torch.tensor([0, 49, -1, -9, 7, 7, -3], dtype=torch.int64)
    ref = torch.stack([a[i] if i >= 0 else b[-i - 1] for i in idx.tolist()])
    o = ops.gather_rows_f32(a.cuda(), b.cuda(), idx.cuda()).cpu()
    ob = ops.gather_rows_f32(a.cuda(), b.cuda(), idx.cuda(), torch.bfloat16).cpu()
    assert torch.equal(o, ref) and torch.equal(ob, ref.bfloat16())
    s = _rand((9, 384), 63).bfloat16()
    i2 = torch.tensor([8, 0, 0, 3, 5], dtype=torch.int64)
    assert torch.equal(ops.gather_rows_bf16(s.cuda(), i2.cuda()).cpu(), s[i2])
    assert torch.equal(ops.cast_bf16(a.cuda()).cpu(), a.bfloat16())


def test_rowdot_softplus(ops):
    x = _rand((11, 1536), 71, 2.0)
    x[3] *= 30                                              # exercises the softplus threshold branch
    w, b = _rand((1536,), 72, 0.05), torch.tensor([0.3])
    ref = F.softplus(x @ w[:, None] + b)
    got = ops.rowdot_softplus(x.cuda(), w.cuda(), b.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------
# GEMM tile configurations at ragged edges (the automatic choice depends on the grid size, so every configuration the
# pipeline can select is also forced here on shapes whose M / N are not multiples of any tile)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [1, 5, 20, 22])
@pytest.mark.parametrize("epi", ["bf16", "res", "f32"])
def test_gemm_forced_tile_configs_ragged(ops, variant, epi):
    M, N, K = 515, 776, 192
    a = _bf(_rand((M, K), 31))
    w = _bf(_rand((N, K), 32, 1.0 / math.sqrt(K)))
    b = _rand((N,), 33, 0.5)
    res = _rand((M, N), 34)
    ref = a @ w.t() + b + (res if epi == "res" else 0)
    code = {"bf16": ops.EPI_BF16, "f32": ops.EPI_F32, "res": ops.EPI_RES_F32}[epi]
    out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), code, residual=res.cuda() if epi == "res" else None,
                   variant=variant)
    torch.cuda.synchronize()
    tol = dict(rtol=2 ** -8, atol=2e-3) if epi == "bf16" else dict(rtol=2e-5, atol=2e-5 * math.sqrt(K))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **tol)


@pytest.mark.parametrize("variant", [1, 20, 22])
def test_gemm_geglu_forced_tile_configs_ragged(ops, variant):
    M, F2, K = 515, 1056, 192
    a = _bf(_rand((M, K), 41))
    w = _bf(_rand((F2, K), 42, 1.0 / math.sqrt(K)))
    b = _rand((F2,), 43, 0.5)
    x, gate = (a @ w.t() + b).chunk(2, dim=-1)
    wp, bp = ops.pack_geglu_rows(w.cuda().bfloat16(), b.cuda())
    out = ops.gemm(a.cuda().bfloat16(), wp, bp, ops.EPI_GEGLU_BF16, variant=variant)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), (x * F.gelu(gate)).numpy(), rtol=2 ** -8, atol=2e-3)


@pytest.mark.parametrize("epi", ["bf16", "res", "geglu"])
def test_gemm_auto_selection_at_batch_size(ops, epi):
    """Grid large enough for the batch-size rules of pick_variant (256x256 tiles; BK = 32 ring for the K = 512 GeGLU),
    ragged in M and N; checked on a row sample against fp32 matmul of the same bf16 operands."""
    M, N, K = 140003, (1056 if epi == "geglu" else 776), 512
    g = torch.Generator().manual_seed(77)
    a = _bf(torch.rand((M, K), generator=g) * 2 - 1)
    w = _bf((torch.rand((N, K), generator=g) * 2 - 1) / math.sqrt(K))
    b = torch.rand((N,), generator=g)
    rows = torch.cat([torch.arange(0, 300), torch.arange(70000, 70300), torch.arange(M - 300, M)])
    h = a[rows] @ w.t() + b
    if epi == "geglu":
        wp, bp = ops.pack_geglu_rows(w.cuda().bfloat16(), b.cuda())
        out = ops.gemm(a.cuda().bfloat16(), wp, bp, ops.EPI_GEGLU_BF16)
        x, gate = h.chunk(2, dim=-1)
        ref = x * F.gelu(gate)
    elif epi == "res":
        res = torch.rand((M, N), generator=g)
        out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), ops.EPI_RES_F32, residual=res.cuda())
        ref = h + res[rows]
    else:
        out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), ops.EPI_BF16)
        ref = h
    torch.cuda.synchronize()
    tol = dict(rtol=2e-5, atol=2e-5 * math.sqrt(K)) if epi == "res" else dict(rtol=2 ** -8, atol=2e-3)
    np.testing.assert_allclose(out[rows.cuda()].float().cpu().numpy(), ref.numpy(), **tol)


@pytest.mark.parametrize("variant", [20, 22])
@pytest.mark.parametrize("K", [64, 128, 192, 1536])
@pytest.mark.parametrize("epi", ["bf16", "res", "gelu_f32"])
def test_gemm_8phase_short_and_odd_k_loops(ops, K, epi, variant):
    """The two-group 256x256 kernel (variant 20) with 1, 2, 3 (odd) and 24 K-tiles: prologue / drain paths of its
    prefetch stream, ragged M and N, every accumulator checked."""
    if variant == 21 and K < 128:
        pytest.skip("the persistent form needs two K-tiles per output tile")
    M, N = 700, 520
    a = _bf(_rand((M, K), 51))
    w = _bf(_rand((N, K), 52, 1.0 / math.sqrt(K)))
    b = _rand((N,), 53, 0.5)
    res = _rand((M, N), 54)
    ref = a @ w.t() + b
    if epi == "res":
        ref = ref + res
    if epi == "gelu_f32":
        ref = F.gelu(ref)
    code = {"bf16": ops.EPI_BF16, "res": ops.EPI_RES_F32, "gelu_f32": ops.EPI_GELU_F32}[epi]
    out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), code, residual=res.cuda() if epi == "res" else None,
                   variant=variant)
    torch.cuda.synchronize()
    tol = dict(rtol=2 ** -8, atol=2e-3) if epi == "bf16" else dict(rtol=2e-5, atol=2e-5 * math.sqrt(K))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **tol)


@pytest.mark.parametrize("variant", [20, 22])
def test_gemm_8phase_race_screen(ops, variant):
    """Exact-integer operands (every product and sum exact in fp32) on a grid of several waves of tiles, repeated: any
    LDS hazard in the staggered two-group schedule (a fragment read before its LDS-DMA landed, a half-tile overwritten
    before its last read) shows up as a wrong integer.  Results must also be identical from launch to launch."""
    M, N, K = 256 * 37 + 19, 256 * 9, 1536
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    ab, wb = a.cuda().bfloat16(), w.cuda().bfloat16()
    ref = (a.cuda() @ w.cuda().t())
    first = None
    for it in range(6):
        out = ops.gemm(ab, wb, None, ops.EPI_F32, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"iteration {it}: {(out != ref).sum().item()} wrong elements"
        first = out if first is None else first
        assert torch.equal(out, first)


# ---------------------------------------------------------------------------------------------
# fp16-operand twins (BASELINE configs[4]: "fp16 with fp32 accumulate"; reference 16-mixed / fp16 flash path)
# ---------------------------------------------------------------------------------------------
def _h(x):
    return x.to(torch.float16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,variant", [(257, 576, 192, 0), (1000, 1536, 512, 0), (513, 4608, 1536, 0), (77, 40, 72, 0),
                                           (700, 520, 192, 20), (515, 776, 192, 1), (300, 192, 192, 5)])
@pytest.mark.parametrize("epi", ["f16", "f32", "res", "gelu_f32"])
def test_gemm_fp16_operands(ops, M, N, K, variant, epi):
    if variant and K % 64:
        pytest.skip("forced tile configurations need K % 64 == 0")
    a = _h(_rand((M, K), 61))
    w = _h(_rand((N, K), 62, 1.0 / math.sqrt(K)))
    b = _rand((N,), 63, 0.5)
    res = _rand((M, N), 64)
    ref = a @ w.t() + b
    if epi == "res":
        ref = ref + res
    if epi == "gelu_f32":
        ref = F.gelu(ref)
    code = {"f16": ops.EPI_BF16, "f32": ops.EPI_F32, "res": ops.EPI_RES_F32, "gelu_f32": ops.EPI_GELU_F32}[epi]
    out = ops.gemm(a.cuda().half(), w.cuda().half(), b.cuda(), code, residual=res.cuda() if epi == "res" else None,
                   variant=variant)
    torch.cuda.synchronize()
    if epi == "f16":
        assert out.dtype == torch.float16
        np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=2 ** -11, atol=3e-4)   # half an fp16 ulp
    else:
        assert out.dtype == torch.float32
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5 * math.sqrt(K))


def test_gemm_geglu_fp16(ops):
    M, F2, K = 515, 1056, 192
    a = _h(_rand((M, K), 71))
    w = _h(_rand((F2, K), 72, 1.0 / math.sqrt(K)))
    b = _rand((F2,), 73, 0.5)
    x, gate = (a @ w.t() + b).chunk(2, dim=-1)
    wp, bp = ops.pack_geglu_rows(w.cuda().half(), b.cuda())
    for variant in (0, 1, 20):
        out = ops.gemm(a.cuda().half(), wp, bp, ops.EPI_GEGLU_BF16, variant=variant)
        torch.cuda.synchronize()
        assert out.dtype == torch.float16
        np.testing.assert_allclose(out.float().cpu().numpy(), (x * F.gelu(gate)).numpy(), rtol=2 ** -10, atol=3e-4)


def test_gemm_mixed_operand_types_are_rejected(ops):
    a = torch.zeros((4, 64), dtype=torch.float16, device="cuda")
    w = torch.zeros((8, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(AssertionError):
        ops.gemm(a, w, None, ops.EPI_F32)


@pytest.mark.parametrize("dh,H,ql,kl,alibi", [ATTN_CASES[i] for i in (1, 2, 3, 4, 5, 6, 9, 12, 13, 16)])
def test_attention_fp16_matches_oracle(ops, dh, H, ql, kl, alibi):
    self_attn = kl is None
    kl = ql if self_attn else kl
    D = H * dh
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32) if alibi else None
    q = _h(_rand((tq, D), 22, 2.0))
    kv = _h(_rand((tk, 2 * D), 23, 2.0))
    k, v = kv[:, :D], kv[:, D:]
    rnd = O.Rounding("fp16")
    ref = torch.zeros(tq, D)
    for b in range(len(ql)):
        a, e, ka, ke = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        ref[a:e] = O.attention(q[a:e].view(-1, H, dh), k[ka:ke].view(-1, H, dh), v[ka:ke].view(-1, H, dh), slopes,
                               rnd).reshape(e - a, D)
    dkv = kv.cuda().half()
    out = ops.attn_varlen(q.cuda().half(), dkv[:, :D], dkv[:, D:], cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh,
                          slopes.cuda() if alibi else None)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16
    # fp16 output (2^-12) + fp16 P rounding at a different running max than the oracle's final max
    np.testing.assert_allclose(out.float().cpu().numpy(), _h(ref).numpy(), rtol=2 ** -9, atol=1.5e-3)


def test_layernorm_cast_pool_fp16_outputs(ops):
    x = _rand((37, 512), 81, 3.0)
    g, b = 1 + 0.1 * _rand((512,), 82), 0.1 * _rand((512,), 83)
    ref = F.layer_norm(x, (512,), g, b, 1e-5)
    out = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), torch.float16)
    assert out.dtype == torch.float16
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=2 ** -11, atol=1e-3)
    c = ops.cast16(x.cuda(), torch.float16)
    assert torch.equal(c.cpu(), x.half())                          # round to nearest even, bit exact
    with ops.compute_dtype(torch.float16):
        assert ops.cast16(x.cuda()).dtype == torch.float16 and ops.layernorm(x.cuda(), g.cuda(), b.cuda()).dtype == torch.float16
    assert ops.cdt() == torch.bfloat16
    cu = torch.tensor([0, 5, 5, 37], dtype=torch.int32)
    m = ops.segment_mean(x.cuda(), cu.cuda(), torch.float16)
    np.testing.assert_allclose(m[0].float().cpu().numpy(), x[:5].mean(0).numpy(), rtol=2 ** -11, atol=1e-3)
    np.testing.assert_allclose(m[2].float().cpu().numpy(), x[5:].mean(0).numpy(), rtol=2 ** -11, atol=1e-3)


# ---------------------------------------------------------------------------------------------
# LayerNorm folded into the GEMMs (vf_gemm_ln_bf16 / vf_ln_finalize / vf_row_stats_cast): the producer epilogue's bf16
# copy and row statistics, the consumer epilogue's correction; shapes chosen so that pick_variant takes each tile
# configuration (64x64: M = 515; 128x128: M = 10854; two-group 256x256: M = 16387, both at N = 1536)
# ---------------------------------------------------------------------------------------------
def _row_sample(M, n=96, seed=5):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(M, generator=g)[:min(n, M)]
    return torch.cat([idx, torch.tensor([0, M - 1])]).unique()


@pytest.mark.parametrize("M,N,K", [(515, 1536, 1536), (10854, 1536, 1024), (16387, 1536, 1536), (40000, 512, 512),
                                   (300, 800, 192), (1, 64, 64)])
@pytest.mark.parametrize("with_res", [True, False])
def test_gemm_ln_producer(ops, M, N, K, with_res):
    a = _bf(_rand((M, K), 301))
    w = _bf(_rand((N, K), 302, 1.0 / math.sqrt(K)))
    b = _rand((N,), 303, 0.5)
    res = _rand((M, N), 304, 3.0) + 0.7 if with_res else None      # non-zero row mean
    s = ops.gemm_ln_producer(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), None if res is None else res.cuda())
    plain = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), ops.EPI_RES_F32 if with_res else ops.EPI_F32,
                     residual=None if res is None else res.cuda())
    torch.cuda.synchronize()
    assert torch.equal(s.x, plain), "the producer epilogue must not change the fp32 result"
    assert torch.equal(s.x16, s.x.bfloat16()), "bf16 copy = round-to-nearest-even of the fp32 stream"
    # the same launch without the fp32 store (a stream that is only read through the next LayerNorm -> Linear pair)
    t = ops.gemm_ln_producer(a.cuda().bfloat16(), w.cuda().bfloat16(), b.cuda(), None if res is None else res.cuda(),
                             need_x=False)
    torch.cuda.synchronize()
    assert t.x is None and torch.equal(t.x16, s.x16) and torch.equal(t.stats, s.stats)
    rows = _row_sample(M)
    x = s.x[rows.cuda()].double().cpu()
    ref = a[rows] @ w.t() + b + (res[rows] if with_res else 0)
    np.testing.assert_allclose(x.float().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5 * math.sqrt(K))
    mean = x.mean(dim=1)
    rstd = 1.0 / torch.sqrt(x.var(dim=1, unbiased=False) + 1e-5)
    st = s.stats[rows.cuda()].double().cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[:, 1].numpy(), rstd.numpy(), rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(515, 1536, 1536), (10854, 1536, 1536), (16387, 1536, 1536), (40000, 512, 512)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_ln_producer_with_16bit_residual(ops, M, N, K, dtype):
    """vf_gemm_ln with residual_dtype = operand type (every tile configuration): the residual is the 16-bit copy of a
    stream (unscaled on read), x = a @ w^T + b + float(res16) / scale.  Bit-identical to the plain fp32-residual GEMM fed
    with that value; the copy and statistics follow from x as in the fp32-residual form; need_x=False drops only the store."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with ops.compute_dtype(td):
        a = _rand((M, K), 341).to(td)
        w = (_rand((N, K), 342, 1.0 / math.sqrt(K))).to(td)
        b = _rand((N,), 343, 0.5)
        res = _rand((M, N), 344, 3.0) + 0.7
        s0 = ops.ln_stream(res.cuda())                                   # the stream whose 16-bit copy is the residual
        scale = ops.x16_scale_for(td)
        assert s0.scale == scale and (scale == 1.0) == (dtype == "bf16")
        assert torch.equal(s0.x16, (res.cuda() * scale).to(td))
        res_val = s0.x16.float() / scale                                  # what the kernel must add
        s = ops.gemm_ln_producer(a.cuda(), w.cuda(), b.cuda(), s0)
        plain = ops.gemm(a.cuda(), w.cuda(), b.cuda(), ops.EPI_RES_F32, residual=res_val.contiguous())
        torch.cuda.synchronize()
        assert torch.equal(s.x, plain)
        assert torch.equal(s.x16, (s.x * scale).to(td))
        t = ops.gemm_ln_producer(a.cuda(), w.cuda(), b.cuda(), s0, need_x=False)
        torch.cuda.synchronize()
        assert t.x is None and torch.equal(t.x16, s.x16) and torch.equal(t.stats, s.stats)
        xd = s.x.double()
        mean = xd.mean(dim=1)
        rstd = 1.0 / torch.sqrt(xd.var(dim=1, unbiased=False) + 1e-5)
        np.testing.assert_allclose(s.stats[:, 0].double().cpu().numpy(), (mean * scale).cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(s.stats[:, 1].double().cpu().numpy(), (rstd / scale).cpu().numpy(), rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(515, 1536, 1024), (10854, 1536, 1024), (16387, 1536, 1024), (40000, 512, 1024), (300, 128, 128)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_ln_producer_with_fp16_trunk(ops, M, N, K, dtype):
    """vf_gemm_ln_t16 (every tile configuration): the layer trunk as a scaled fp16 copy whatever the operand type.
    x = a @ w^T + b + float(t16_in) / T16_SCALE must be bit-identical to the plain fp32-residual GEMM fed with that value;
    the operand-type copy and the statistics follow from x as in every producer; t16_out = fp16(x * T16_SCALE), round to
    nearest even; need_x=False / need_t16=False drop only their stores."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    with ops.compute_dtype(td):
        a = _rand((M, K), 361).to(td)
        w = (_rand((N, K), 362, 1.0 / math.sqrt(K))).to(td)
        b = _rand((N,), 363, 0.5)
        res = (_rand((M, N), 364, 3.0) + 0.7) * (10.0 ** _rand((M, 1), 365, 1.5))     # rows over three decades
        t_in = ops.trunk16_of(res.cuda())
        assert t_in.dtype == torch.float16 and torch.equal(t_in, (res.cuda() * ops.T16_SCALE).half())
        res_val = t_in.float() / ops.T16_SCALE                                       # what the kernel must add
        s = ops.gemm_ln_producer(a.cuda(), w.cuda(), b.cuda(), None, trunk16=t_in, need_t16=True)
        plain = ops.gemm(a.cuda(), w.cuda(), b.cuda(), ops.EPI_RES_F32, residual=res_val.contiguous())
        torch.cuda.synchronize()
        scale = ops.x16_scale_for(td)
        assert torch.equal(s.x, plain)
        assert torch.equal(s.x16, (s.x * scale).to(td))
        assert s.t16.dtype == torch.float16 and torch.equal(s.t16, (s.x * ops.T16_SCALE).half())
        t = ops.gemm_ln_producer(a.cuda(), w.cuda(), b.cuda(), None, need_x=False, trunk16=t_in, need_t16=True)
        u = ops.gemm_ln_producer(a.cuda(), w.cuda(), b.cuda(), None, need_x=True, trunk16=t_in, need_t16=False)
        torch.cuda.synchronize()
        assert t.x is None and torch.equal(t.x16, s.x16) and torch.equal(t.stats, s.stats) and torch.equal(t.t16, s.t16)
        assert u.t16 is None and torch.equal(u.x, s.x) and torch.equal(u.x16, s.x16) and torch.equal(u.stats, s.stats)
        xd = s.x.double()
        mean = xd.mean(dim=1)
        rstd = 1.0 / torch.sqrt(xd.var(dim=1, unbiased=False) + 1e-5)
        np.testing.assert_allclose(s.stats[:, 0].double().cpu().numpy(), (mean * scale).cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(s.stats[:, 1].double().cpu().numpy(), (rstd / scale).cpu().numpy(), rtol=2e-5)
        rows = torch.tensor([0, M // 2, M - 1])
        sr = ops.ln_stream_rows(t, rows.cuda())
        assert torch.equal(sr.t16, s.t16[rows.cuda()])


@pytest.mark.parametrize("M,N,K", [(515, 4608, 1536), (16387, 1536, 512), (10854, 2048, 1536)])
def test_gemm_ln_fp16_consumer_matches_folded_oracle_and_unfolded_pair(ops, M, N, K):
    """The LayerNorm fold for fp16 operands (BASELINE configs[4]): the stream copy is fp16(x * 2^-4) and the statistics are
    (mean * c, rstd / c); the consumer must reproduce the oracle's folded statement (rounding="fp16": the power-of-two
    scale rounds identically inside the fp16 normal range) on every row, and stay at fp16 level from the unfolded pair.
    Rows carry a large common offset so that an unscaled fp16 copy would not be at risk here but the scale path is
    exercised with values over three decades."""
    from variantformer_amd.seq2gene.modules.layers import packed_linear_ln
    geglu = N == 2048
    x = (_rand((M, K), 351, 2.0) + _rand((M, 1), 352, 1.5)) * (10.0 ** _rand((M, 1), 353, 1.5))      # row scales 0.03 .. 30
    lin, norm = torch.nn.Linear(K, N), torch.nn.LayerNorm(K)
    with torch.no_grad():
        lin.weight.copy_(_rand((N, K), 354, 1.0 / math.sqrt(K)))
        lin.bias.copy_(_rand((N,), 355, 0.5))
        norm.weight.copy_(1.0 + _rand((K,), 356, 0.3))
        norm.bias.copy_(_rand((K,), 357, 0.2))
    rnd = O.Rounding("fp16", fold_ln=True)
    ref = O.linear(rnd.ln(x, norm.weight.detach(), norm.bias.detach()), lin.weight.detach(), lin.bias.detach(), rnd)
    plain = F.linear(F.layer_norm(x, (K,), norm.weight, norm.bias, 1e-5), lin.weight, lin.bias).detach()
    if geglu:
        ref = ref[:, :N // 2] * F.gelu(ref[:, N // 2:])
        plain = plain[:, :N // 2] * F.gelu(plain[:, N // 2:])
    lin, norm = lin.cuda(), norm.cuda()
    with ops.compute_dtype(torch.float16):
        wp, bp, cs = packed_linear_ln(lin, norm, geglu=geglu)
        assert wp.dtype == torch.float16
        s = ops.ln_stream(x.cuda())
        assert s.x16.dtype == torch.float16 and s.scale == 2.0 ** -4
        out = ops.gemm_ln_consumer(s, wp, bp, cs, ops.EPI_GEGLU_BF16 if geglu else ops.EPI_BF16)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16
    got = out.float().cpu()
    np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=2 ** -10, atol=1e-3)       # all rows
    assert _rel_err(got, plain) < 4e-3


@pytest.mark.parametrize("rows,D", [(1, 64), (37, 512), (1000, 1536), (5, 4096)])
def test_ln_stream_stats_and_copy(ops, rows, D):
    x = _rand((rows, D), 311, 2.0) + _rand((rows, 1), 312, 4.0)
    s = ops.ln_stream(x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(s.x16.cpu(), x.bfloat16())
    xd = x.double()
    np.testing.assert_allclose(s.stats[:, 0].double().cpu().numpy(), xd.mean(dim=1).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(s.stats[:, 1].double().cpu().numpy(),
                               (1.0 / torch.sqrt(xd.var(dim=1, unbiased=False) + 1e-5)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(515, 4608, 1536), (10854, 1536, 1536), (16387, 4608, 1536), (40000, 1536, 512),
                                   (77, 72, 64)])
@pytest.mark.parametrize("geglu", [False, True])
def test_gemm_ln_consumer_matches_folded_oracle(ops, M, N, K, geglu):
    """LayerNorm -> Linear as the HIP path evaluates it against the oracle's statement of the same rounding points
    (oracle.linear on a pending LayerNorm), and against the unfolded fp32 pair at bf16-level tolerance."""
    from variantformer_amd.seq2gene.modules.layers import packed_linear_ln
    if geglu and N % 32:
        pytest.skip("GEGLU packing needs 2F % 32 == 0")
    x = _rand((M, K), 321, 2.0) + _rand((M, 1), 322, 1.5) + 0.3 * _rand((1, K), 323, 4.0)
    lin = torch.nn.Linear(K, N)
    norm = torch.nn.LayerNorm(K)
    with torch.no_grad():
        lin.weight.copy_(_rand((N, K), 324, 1.0 / math.sqrt(K)))
        lin.bias.copy_(_rand((N,), 325, 0.5))
        norm.weight.copy_(1.0 + _rand((K,), 326, 0.3))
        norm.bias.copy_(_rand((K,), 327, 0.2))
    rows = _row_sample(M)
    rnd = O.Rounding("bf16", fold_ln=True)
    ref = O.linear(rnd.ln(x[rows], norm.weight.detach(), norm.bias.detach()), lin.weight.detach(), lin.bias.detach(), rnd)
    plain = F.linear(F.layer_norm(x[rows], (K,), norm.weight, norm.bias, 1e-5), lin.weight, lin.bias).detach()
    if geglu:
        ref = ref[:, :N // 2] * F.gelu(ref[:, N // 2:])
        plain = plain[:, :N // 2] * F.gelu(plain[:, N // 2:])
    lin, norm = lin.cuda(), norm.cuda()
    wp, bp, cs = packed_linear_ln(lin, norm, geglu=geglu)
    s = ops.ln_stream(x.cuda())
    out = ops.gemm_ln_consumer(s, wp, bp, cs, ops.EPI_GEGLU_BF16 if geglu else ops.EPI_BF16)
    torch.cuda.synchronize()
    got = out[rows.cuda()].float().cpu()
    # same rounding points: differences = fp32 summation order + one bf16 output rounding
    np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=2 ** -7, atol=4e-3)
    assert _rel_err(got, plain) < 2e-2


@pytest.mark.parametrize("M,N,K", [(16387, 4608, 1536), (40000, 1536, 512), (16387, 2048, 1536), (40000, 2048, 512)])
def test_gemm_ln_consumer_every_row_vs_separate_layernorm_on_gpu(ops, M, N, K):
    """EVERY row and column of the folded LayerNorm -> Linear (gemm8x_kernel<.., LN consumer>, persistent 256x256 tiles at
    these grids) against the unfolded pair on the same GPU: vf_layernorm (bf16 out) followed by vf_gemm_bf16.  The two
    round at different points (bf16(x), bf16(gamma.W) vs bf16(LN(x)), bf16(W)), so they agree at bf16 level only -- but a
    tile, a row statistic or a colsum applied to the wrong rows is an O(1) error somewhere, and no row is left unsampled.
    N = 2048 is the GeGLU form (interleaved rows, x * gelu(gate) epilogue)."""
    from variantformer_amd.seq2gene.modules.layers import packed_linear, packed_linear_ln
    geglu = N == 2048
    x = _rand((M, K), 331, 2.0) + _rand((M, 1), 332, 1.5) + 0.3 * _rand((1, K), 333, 4.0)
    lin, norm = torch.nn.Linear(K, N), torch.nn.LayerNorm(K)
    with torch.no_grad():
        lin.weight.copy_(_rand((N, K), 334, 1.0 / math.sqrt(K)))
        lin.bias.copy_(_rand((N,), 335, 0.5))
        norm.weight.copy_(1.0 + _rand((K,), 336, 0.3))
        norm.bias.copy_(_rand((K,), 337, 0.2))
    lin, norm = lin.cuda(), norm.cuda()
    xc = x.cuda()
    epi = ops.EPI_GEGLU_BF16 if geglu else ops.EPI_BF16
    wp, bp, cs = packed_linear_ln(lin, norm, geglu=geglu)
    folded = ops.gemm_ln_consumer(ops.ln_stream(xc), wp, bp, cs, epi).float()
    w, b = packed_linear(lin, geglu=geglu)
    plain = ops.gemm(ops.layernorm(xc, norm.weight, norm.bias, torch.bfloat16), w, b, epi).float()
    torch.cuda.synchronize()
    assert folded.shape == plain.shape == (M, N // 2 if geglu else N)
    diff = (folded - plain).abs()
    bound = 3e-2 + 2 ** -6 * plain.abs()               # two independent bf16-operand evaluations of an O(1) dot product
    bad = diff > bound
    assert not bool(bad.any()), f"{int(bad.sum())} elements off, worst {float(diff.max()):.3e} at {torch.nonzero(bad)[:4].tolist()}"
    assert float(diff.mean()) < 4e-3                    # and no systematic offset: the mean difference is rounding noise
    # row-wise: the worst ROW (mean |diff| over its columns) is not an outlier against the typical row
    per_row = diff.mean(dim=1)
    assert float(per_row.max()) < 6 * float(per_row.median()) + 1e-3, "one row stands out: wrong statistics for that row?"


@pytest.mark.parametrize("M,N,K", [(256 * 37 + 19, 256 * 9, 1536), (40000, 1536, 512)])
def test_gemm_ln_consumer_exact_integer_race_screen(ops, M, N, K):
    """Race screen of the LayerNorm-consumer epilogue of the persistent 256x256 kernel (bias', colsum and the tile's 256
    (mean, rstd) pairs arrive by LDS-DMA in a side area, double-buffered across output tiles): operands, statistics and
    colsum are small integers / powers of two, so (acc - mean * colsum) * rstd + bias is exact in fp32 and its bf16
    rounding is unique.  Every element must equal the host-side formula, on every repeat."""
    g = torch.Generator().manual_seed(11)
    a = torch.randint(-1, 2, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float()
    mean = torch.randint(-2, 3, (M,), generator=g).float()
    rstd = torch.tensor([0.25, 0.5, 1.0, 2.0])[torch.randint(0, 4, (M,), generator=g)]
    bias = torch.randint(-8, 9, (N,), generator=g).float()
    colsum = w.sum(dim=1)
    stats = torch.stack([mean, rstd], dim=1).contiguous().cuda()
    ab, wb = a.cuda().bfloat16(), w.cuda().bfloat16()
    acc = ops.gemm(ab, wb, None, ops.EPI_F32, variant=20)          # exact integers (screened by test_gemm_8phase_race_screen)
    assert torch.equal(acc[:64].cpu(), a[:64] @ w.t())
    ref = ((acc - mean.cuda()[:, None] * colsum.cuda()[None, :]) * rstd.cuda()[:, None] + bias.cuda()[None, :]).bfloat16()
    s = ops.LnStream(None, ab, stats)
    for it in range(5):
        out = ops.gemm_ln_consumer(s, wb, bias.cuda(), colsum.cuda(), ops.EPI_BF16)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"iteration {it}: {(out != ref).sum().item()} wrong elements"


@pytest.mark.parametrize("M,N,K", [(256 * 37 + 19, 1536, 1536), (40000, 512, 512), (256 * 20 + 3, 1536, 1024)])
@pytest.mark.parametrize("with_res", [True, False])
def test_gemm_ln_producer_exact_integer_race_screen(ops, M, N, K, with_res):
    """Race screen of the LayerNorm-producer epilogue (gemm8_kernel<.., LN producer>: fp32 store, bf16 copy, per-part
    (sum, second moment about the part mean) by DPP): integer operands and residuals keep x and sum(x) exact in fp32, so the fp32
    stream, its bf16 copy and the partial statistics have unique values; all rows, repeated."""
    g = torch.Generator().manual_seed(13)
    a = torch.randint(-1, 2, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    res = torch.randint(-16, 17, (M, N), generator=g).float() if with_res else None
    ab, wb = a.cuda().bfloat16(), w.cuda().bfloat16()
    x_ref = ops.gemm(ab, wb, None, ops.EPI_F32, variant=20) + bias.cuda()[None, :]
    if with_res:
        x_ref = x_ref + res.cuda()
    assert float(x_ref.abs().max()) < 256 and float((x_ref * x_ref).sum(dim=1).max()) < 2 ** 24     # exactness premise
    xd = x_ref.double()
    mean_ref = xd.mean(dim=1)
    rstd_ref = 1.0 / torch.sqrt((xd * xd).mean(dim=1) - mean_ref * mean_ref + 1e-5)
    first = None
    for it in range(4):
        s = ops.gemm_ln_producer(ab, wb, bias.cuda(), None if res is None else res.cuda())
        torch.cuda.synchronize()
        assert torch.equal(s.x, x_ref), f"iteration {it}: {(s.x != x_ref).sum().item()} wrong fp32 elements"
        assert torch.equal(s.x16, x_ref.bfloat16())
        np.testing.assert_allclose(s.stats[:, 0].double().cpu().numpy(), mean_ref.cpu().numpy(), rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(s.stats[:, 1].double().cpu().numpy(), rstd_ref.cpu().numpy(), rtol=4e-6)
        first = s.stats.clone() if first is None else first
        assert torch.equal(s.stats, first)


@pytest.mark.parametrize("ratio", [1.0, 10.0, 100.0])
def test_ln_fold_rows_with_large_mean(ops, ratio, monkeypatch):
    """Rows whose mean is `ratio` times their standard deviation (round-2 advice: the synthetic-weight goldens never visit
    that regime, a real checkpoint's residual stream may).  (1) The row statistics of the producer path (partial sums
    merged with the parallel-variance formula) stay at fp32 accuracy whatever the mean: E[x^2] - mean^2 would lose
    log10(ratio^2) digits.  (2) The folded LayerNorm -> Linear reads bf16(x), not bf16(LN(x)): its rounding noise grows
    with |mean| / std.  This test states that growth: error vs the exact fp32 pair <= 2^-8 * (1.5 + ratio) on an O(1)
    output, while the separate-LayerNorm path (VF_LN_FOLD=0) stays at 2^-7 -- so the tolerance of the default path on a
    stream with a large row mean is known before a real checkpoint runs."""
    from variantformer_amd.seq2gene.modules.layers import packed_linear, packed_linear_ln
    M, K, N = 4099, 1536, 1536
    ops.ln_fold_alert(torch.device("cuda", torch.cuda.current_device()))         # clear whatever earlier tests left
    std = 0.5
    x_c = _rand((M, K), 401, std * math.sqrt(3.0))                       # uniform with standard deviation `std`
    x_c = x_c - x_c.mean(dim=1, keepdim=True)
    mu = ratio * std * (1.0 + 0.2 * _rand((M, 1), 402))
    # the stream as a producer GEMM writes it: x = a @ w^T + residual with an exactly representable product part
    a = torch.zeros((M, 64)); w = torch.zeros((K, 64))
    res = (x_c + mu).contiguous()
    s = ops.gemm_ln_producer(a.cuda().bfloat16(), w.cuda().bfloat16(), None, res.cuda())
    torch.cuda.synchronize()
    xd = res.double()
    mean_ref = xd.mean(dim=1)
    rstd_ref = 1.0 / torch.sqrt(xd.var(dim=1, unbiased=False) + 1e-5)
    np.testing.assert_allclose(s.stats[:, 0].double().cpu().numpy(), mean_ref.numpy(), rtol=2e-6)
    np.testing.assert_allclose(s.stats[:, 1].double().cpu().numpy(), rstd_ref.numpy(), rtol=1e-5)      # at every ratio
    lin, norm = torch.nn.Linear(K, N), torch.nn.LayerNorm(K)
    with torch.no_grad():
        lin.weight.copy_(_rand((N, K), 403, 1.0 / math.sqrt(K)))
        lin.bias.copy_(_rand((N,), 404, 0.5))
        norm.weight.copy_(1.0 + _rand((K,), 405, 0.3))
        norm.bias.copy_(_rand((K,), 406, 0.2))
    exact = F.linear(F.layer_norm(res.double(), (K,), norm.weight.double(), norm.bias.double(), 1e-5), lin.weight.double(),
                     lin.bias.double())
    lin, norm = lin.cuda(), norm.cuda()
    wp, bp, cs = packed_linear_ln(lin, norm)
    folded = ops.gemm_ln_consumer(s, wp, bp, cs, ops.EPI_BF16).double().cpu()
    wq, bq = packed_linear(lin)
    plain = ops.gemm(ops.layernorm(res.cuda(), norm.weight, norm.bias, torch.bfloat16), wq, bq, ops.EPI_BF16).double().cpu()
    e_fold, e_plain = float((folded - exact.detach()).abs().max()), float((plain - exact.detach()).abs().max())
    print(f"[ln fold, |mean| = {ratio:g} x std] max abs error on an O(1) output: folded {e_fold:.3e}, separate LayerNorm {e_plain:.3e}")
    assert e_plain < 2 ** -6
    assert e_fold < 2 ** -8 * (3.0 + 1.5 * ratio)          # measured 1.3e-2 / 5.6e-2 / 4.6e-1 at ratio 1 / 10 / 100
    # the statistics kernels flag rows beyond LN_FOLD_RATIO_LIMIT (8) standard deviations
    assert ops.ln_fold_alert(torch.device("cuda", torch.cuda.current_device())) == (ratio > ops.LN_FOLD_RATIO_LIMIT)


def test_gelu_epilogue_accuracy_over_the_whole_range(ops):
    """The kernels' erf GELU (erfc-based, vf_common.h gelu_erf4) against float64 erf on a dense grid of bf16-exact
    inputs in [-12, 12] pushed through an identity GEMM with the fp32 GELU epilogue: absolute error <= 5e-7 (what
    0.5 x (1 + erff(x / sqrt 2)) gives in fp32 as well), and no blow-up of the relative error on the negative tail."""
    K = 64
    xs = torch.linspace(-12, 12, 64 * 4096).bfloat16().float().view(-1, K)
    eye = torch.eye(K)
    out = ops.gemm(xs.cuda().bfloat16(), eye.cuda().bfloat16(), None, ops.EPI_GELU_F32).cpu().double()
    xd = xs.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    assert float((out - ref).abs().max()) <= 5e-7
    tail = (xd < -1) & (xd > -5)
    assert float(((out - ref).abs() / ref.abs())[tail].max()) < 5e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("d", [512, 1024, 64, 192, 2048])
def test_segment_mean16_vs_float64(ops, dtype, d):
    """vf_segment_mean16: per-window mean of a 16-bit stream (strided rows), fp32 result and the [hi | lo] split.  Ragged windows
    incl. 1 row, 3 rows (fewer than the rows in flight), 200 rows and an empty one (NaN, the reference's 0 / 0); against float64
    on the same 16-bit values: fp32 to 2e-6 relative of the row scale, hi + lo to 2^-15 (two 16-bit mantissas)."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    lens = [200, 1, 3, 0, 64, 65, 7, 130, 97, 2]
    n = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
    big = (_rand((n, d + 16), 81, 3.0)).cuda().to(td)
    x = big[:, 8:8 + d]                                   # a strided view: rows d + 16 apart
    scale = 16.0
    f = ops.segment_mean16(x, cu, in_scale=scale)
    sp = ops.segment_mean16(x, cu, in_scale=scale, split=True)
    torch.cuda.synchronize()
    assert f.shape == (len(lens), d) and sp.shape == (len(lens), 2 * d) and sp.dtype == td
    xd = x.double().cpu()
    for w, ln in enumerate(lens):
        a = int(cu[w])
        if ln == 0:
            assert torch.isnan(f[w]).all() and torch.isnan(sp[w].float()).all()
            continue
        ref = xd[a:a + ln].mean(dim=0) * scale
        tol = float(ref.abs().max())
        assert float((f[w].double().cpu() - ref).abs().max()) <= 2e-6 * tol + 1e-30
        rec = sp[w, :d].double().cpu() + sp[w, d:].double().cpu()
        assert float((rec - ref).abs().max()) <= 2 ** -15 * tol


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("geom", ["seq2reg_windows", "seq2reg_chunks", "gene_self"])
def test_attention_row_map_form_is_bit_identical_to_gather_then_attend(ops, dtype, geom):
    """vf_attn_varlen_fwd_rows (ABI 8): q / k / v are tables of distinct projected rows and the kernel gathers in its loads
    (the first layers' projection by lookup).  Bit for bit the plain entry on the gathered rows, for the three geometries
    that have a row-map kernel (ragged sequences, a one-token sequence, duplicate and out-of-order table rows); any other
    geometry is reported unsupported and rejected by name."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rng = np.random.default_rng(11)
    if geom == "seq2reg_windows":
        dh, H, alibi, lens = 64, 8, False, [int(x) for x in rng.integers(1, 129, 160)] + [128, 1]
    elif geom == "seq2reg_chunks":
        dh, H, alibi, lens = 64, 8, False, [int(x) for x in rng.integers(100, 201, 140)] + [200, 129, 7]
    else:
        dh, H, alibi, lens = 48, 32, True, [201, 201, 130, 37, 201, 1, 220]      # (a 256-key image leaves no room for 3 blocks: no row-map kernel)
    D = H * dh
    n_tab = 5000
    tab = (_rand((n_tab, 3 * D), 71, 1.5)).cuda().to(td)
    T = sum(lens)
    rows = torch.from_numpy(rng.integers(0, n_tab, T)).long()
    rows[:7] = rows[0]                                   # duplicates
    rows = rows.cuda().contiguous()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
    slopes = torch.tensor([2.0 ** (-(i + 1) / 4) for i in range(H)], dtype=torch.float32).cuda() if alibi else None
    assert ops.attn_rows_supported(dh, alibi, len(lens), H, max(lens), max(lens), True)
    got = ops.attn_varlen(tab[:, :D], tab[:, D:2 * D], tab[:, 2 * D:], cu, None, max(lens), max(lens), H, dh, slopes,
                          q_log2=True, rows=rows)
    g = ops.gather_rows_bf16(tab, rows)
    want = ops.attn_varlen(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], cu, None, max(lens), max(lens), H, dh, slopes, q_log2=True)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (T, D) and torch.isfinite(got.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # geometries without a row-map kernel: reported, and rejected by the entry itself
    assert not ops.attn_rows_supported(dh, alibi, len(lens), H, 300, 300, True)
    assert not ops.attn_rows_supported(dh, alibi, len(lens), H, max(lens), max(lens), False)
    assert not ops.attn_rows_supported(32, alibi, len(lens), H, max(lens), max(lens), True)
    from variantformer_amd._lib import VFError
    with pytest.raises(VFError, match="row-map"):
        ops.attn_varlen(tab[:, :D], tab[:, D:2 * D], tab[:, 2 * D:], cu, None, max(lens), max(lens), H, dh, slopes,
                        q_log2=False, rows=rows)
    if geom == "seq2reg_windows":
        # round-5 advice: <= 128 queries against 129-256 keys is served by the tiled kernel, which reads no row map; the
        # query says so and the entry refuses instead of attending over the wrong rows (max_k != max_q is reachable through
        # the public entry, not through the model)
        assert not ops.attn_rows_supported(64, False, len(lens), H, 128, 200, True)
        with pytest.raises(VFError, match="row-map"):
            ops.attn_varlen(tab[:, :D], tab[:, D:2 * D], tab[:, 2 * D:], cu, None, 128, 200, H, dh, None, q_log2=True, rows=rows)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("dh,H", [(48, 32), (64, 8), (32, 4)])
def test_attention_counted_keys_matches_oracle_and_the_expanded_form(ops, dtype, dh, H):
    """vf_attn_counted_keys (ABI 9): cross attention against keys that are copies of <= 16 distinct rows = softmax over the distinct
    rows with log2(count) added.  Every element against oracle.attention_counted on the same rounded operands (one rounding of
    the output: <= 1 ulp), and against the HIP attention over the EXPANDED keys (which rounds P to 16 bits: close, not equal);
    ragged sequences, a one-query sequence, an empty one, labels that are absent from a sequence."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    D, C = H * dh, 9
    rng = np.random.default_rng(3)
    lens = [300, 1, 0, 77, 1024, 5]
    tq = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    labels = torch.from_numpy(rng.integers(0, C, tq)).long()
    labels[cu[3]:cu[4]] = 4                                             # a sequence that holds ONE label
    q = rnd.r(_rand((tq, D), 81, 1.2))
    tab = rnd.r(_rand((C, 2 * D), 82, 1.5))
    cnt = torch.zeros(len(lens), C)
    for b in range(len(lens)):
        cnt[b] = torch.bincount(labels[cu[b]:cu[b + 1]], minlength=C).float()
    got = ops.attn_counted_keys(q.cuda().to(td), tab.cuda().to(td), torch.log2(cnt).cuda().contiguous(), cu.cuda(), max(lens), H, dh)
    kv = tab[labels]                                                    # the expanded keys the reference's form attends over
    exp = ops.attn_varlen(q.cuda().to(td), kv[:, :D].cuda().to(td).contiguous(), kv[:, D:].cuda().to(td).contiguous(), cu.cuda(),
                          cu.cuda(), max(lens), max(lens), H, dh, q_log2=True)
    torch.cuda.synchronize()
    got, exp = got.float().cpu(), exp.float().cpu()
    ref = torch.zeros(tq, D)
    for b in range(len(lens)):
        a, e = int(cu[b]), int(cu[b + 1])
        if e > a:
            present = [c for c in range(C) if cnt[b, c] > 0]
            ref[a:e] = O.attention_counted(q[a:e].view(-1, H, dh), tab[present, :D].view(-1, H, dh), tab[present, D:].view(-1, H, dh),
                                           cnt[b, present], True).reshape(e - a, D)
    assert torch.isfinite(got).all()
    ulp = 2 ** -7 if dtype == "bf16" else 2 ** -10           # one unit in the last place, relative to the smaller neighbour
    np.testing.assert_allclose(got.numpy(), rnd.r(ref).numpy(), rtol=ulp, atol=ulp * 1e-2)
    # ... and only where fp32 summation order decides a tie (measured: 2e-5 of the elements in bf16, 1.3e-3 in fp16)
    assert float((got != rnd.r(ref)).float().mean()) < (1e-3 if dtype == "bf16" else 1e-2)
    # (the expanded form rounds P to 16 bits before P . V: 16-bit-level agreement, measured max |diff| 4.4e-3 in bf16)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=2 ** -6 if dtype == "bf16" else 2 ** -9, atol=1e-2 if dtype == "bf16" else 2e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_lowrank_context_attention_pieces(ops, dtype):
    """The two new pieces of the low-rank context cross attention (ABI 10): the LayerNorm-consumer GEMM with fp32 OUTPUT
    (logits) against its 16-bit-output twin and the folded oracle, and vf_softmax_counted (16-bit softmax over <= 16 distinct
    keys with log2 counts, padding slots zero) element by element."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    rnd = O.Rounding(dtype)
    M, N, K = 5000, 320, 1536
    x = _rand((M, K), 91, 1.0) + 0.3
    w = _rand((N, K), 92, 0.05)
    b = _rand((N,), 93, 0.1)
    gamma, beta = 1.0 + 0.2 * _rand((K,), 94, 1.0), 0.1 * _rand((K,), 95, 1.0)
    with ops.compute_dtype(td):
        s = ops.ln_stream(x.cuda())
        wg = ops.cast16((w * gamma[None, :]).cuda().contiguous())
        bias = (w @ beta + b).cuda().contiguous()
        cs = wg.float().sum(dim=1).contiguous()
        f32 = ops.gemm_ln_consumer(s, wg, bias, cs, ops.EPI_F32)
        b16 = ops.gemm_ln_consumer(s, wg, bias, cs, ops.EPI_BF16)
    torch.cuda.synchronize()
    assert f32.dtype == torch.float32 and f32.shape == (M, N)
    assert torch.equal(f32.to(td), b16)                                   # the same accumulator, rounded or not
    want = O.linear(O.LnPending(x, gamma, beta), w, b, rnd) if dtype == "bf16" else None
    if want is not None:
        np.testing.assert_allclose(f32.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    # softmax over counted keys
    H, Cp, C = 32, 10, 9
    lens = [700, 1, 0, 33, 1300]
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    sc = _rand((T, H * Cp), 96, 4.0)
    cnt = torch.from_numpy(np.random.default_rng(2).integers(0, 50, (len(lens), C))).float()
    cnt[1] = 0; cnt[1, 4] = 1                                              # one label only
    cnt[3, 0] = 0                                                          # an absent label
    got = ops.softmax_counted(sc.cuda(), torch.log2(cnt).cuda().contiguous(), cu.cuda(), max(lens), H, Cp, out_dtype=td).float().cpu()
    ref = torch.zeros(T, H, Cp)
    for bb in range(len(lens)):
        a, e = int(cu[bb]), int(cu[bb + 1])
        if e > a:
            t = sc[a:e].view(-1, H, Cp)[:, :, :C] + torch.log2(cnt[bb])[None, None, :]
            p = torch.exp2(t - t.max(dim=-1, keepdim=True).values)
            ref[a:e, :, :C] = p / p.sum(dim=-1, keepdim=True)
    ulp = 2 ** -7 if dtype == "bf16" else 2 ** -10
    np.testing.assert_allclose(got.numpy(), rnd.r(ref.view(T, -1)).numpy(), rtol=ulp, atol=1e-7)
    assert float(got.view(T, H, Cp)[:, :, C:].abs().max()) == 0.0         # padding slots
