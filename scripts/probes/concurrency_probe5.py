"""Round 6, concurrency probe 5: which OTHER kernels go wrong beside a co-resident kernel of another kind?  (Probe 4: the
LayerNorm-consumer epilogue of gemm_mfma_kernel -- packed-fp32 v_pk_fma_f32 -- loses its accumulator term in lanes 48..63 beside
an attention kernel; the same library built without packed-fp32 instructions does not.)  Victims on a side stream, co-runners on
the main stream, every output compared bit for bit with the one computed alone.
    python scripts/probes/concurrency_probe5.py          (or through scripts/probes/with_lib.py <lib>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from variantformer_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
def rnd(*shape, scale=1.0):
    return (torch.rand(shape, device=dev, generator=g) * 2 - 1) * scale
def cu(lens):
    return torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)

G = 8
K = 1536
# seq2reg windows: 8 heads x 64, 128-token windows
nw, wl = 1024 * G, 128
qkv_w = rnd(nw * wl // 8, 3 * 512, scale=0.5).bfloat16()
cu_w = cu([wl] * (nw // 8))
# gene self attention: 32 heads x 48, 10854 tokens a gene
ql = [54 * 201] * 2
qkv_g = rnd(sum(ql), 3 * K, scale=0.35).bfloat16()
cu_g = cu(ql)
x32 = rnd(1024 * G, K)
gam, bet = rnd(K), rnd(K)
a64, w64, b64 = rnd(1728, K).bfloat16(), (rnd(1536, K) / K ** 0.5).bfloat16(), rnd(1536)
a1, w1, b1 = rnd(1024 * G, K).bfloat16(), (rnd(4608, K) / K ** 0.5).bfloat16(), rnd(4608)
res = rnd(1024 * G, 1536)
wd = (rnd(1536, K) / K ** 0.5).bfloat16()
bd = rnd(1536)
s1 = ops.ln_stream(rnd(1024 * G, K))
c1 = rnd(4608)


def tup(x):
    if isinstance(x, torch.Tensor):
        return (x,)
    if isinstance(x, ops.LnStream):
        return tuple(t for t in (x.x, x.x16, x.stats, x.t16) if t is not None)
    return tuple(t for y in x for t in tup(y))


victims = {
    "seq2reg window attention (one block per window)": lambda: ops.attn_varlen(qkv_w[:, :512], qkv_w[:, 512:1024], qkv_w[:, 1024:], cu_w, cu_w, wl, wl, 8, 64, None, q_log2=True),
    "gene self attention (32x32x16 kernel)": lambda: ops.attn_varlen(qkv_g[:, :K], qkv_g[:, K:2 * K], qkv_g[:, 2 * K:], cu_g, cu_g, max(ql), max(ql), 32, 48, None, q_log2=True),
    "layernorm kernel (fp32 -> 16 bit)": lambda: ops.layernorm(x32, gam, bet),
    "row statistics + 16-bit copy (ln_stream)": lambda: ops.ln_stream(x32),
    "plain GEMM 64x64 tiles, GELU": lambda: ops.gemm(a64, w64, b64, ops.EPI_GELU_BF16),
    "LayerNorm producer GEMM (+ fp32 residual)": lambda: ops.gemm_ln_producer(a1, wd, bd, res),
    "LayerNorm consumer GEMM 128x128": lambda: ops.gemm_ln_consumer(s1, w1, b1, c1, ops.EPI_BF16),
}
H, dh = 32, 48
D = H * dh
xq, xk = [54 * 201] * G, [1024] * G
cq, ck = cu(xq), cu(xk)
q, kv = rnd(sum(xq), D, scale=0.35).bfloat16(), rnd(sum(xk), 2 * D).bfloat16()
a2, w2, b2 = rnd(54 * 201 * G, K).bfloat16(), (rnd(1536, K) / K ** 0.5).bfloat16(), rnd(1536)
s2 = ops.ln_stream(rnd(54 * 201 * G, K))
c2 = rnd(1536)
corunners = {
    "cross attention (32x32x16 kernel)": lambda: ops.attn_varlen(q, kv[:, :D], kv[:, D:], cq, ck, max(xq), max(xk), H, dh, None, q_log2=True),
    "plain GEMM 128x128 (variant 1)": lambda: ops.gemm(a2, w2, b2, ops.EPI_BF16, variant=1),
    "LayerNorm consumer GEMM 128x128": lambda: ops.gemm_ln_consumer(s2, w2, b2, c2, ops.EPI_BF16),
    "seq2reg window attention": victims["seq2reg window attention (one block per window)"],
}
with torch.no_grad():
    refs = {n: [t.clone() for t in tup(f())] for n, f in victims.items()}
    corefs = {n: [t.clone() for t in tup(f())] for n, f in corunners.items()}
    kern = {}
    for n, f in victims.items():
        f()
        kern[n] = ops.last_kernel("attn" if "attention" in n else "gemm") if ("GEMM" in n or "attention" in n) else "-"
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for cn, co in corunners.items():
        for vn, vf in victims.items():
            bad = cobad = 0
            for rep in range(5):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    outs = [tup(vf()) for _ in range(3)]
                couts = [tup(co()) for _ in range(2)]
                main.wait_stream(side)
                torch.cuda.synchronize()
                bad += sum(0 if all(torch.equal(x, y) for x, y in zip(o, refs[vn])) else 1 for o in outs)
                cobad += sum(0 if all(torch.equal(x, y) for x, y in zip(o, corefs[cn])) else 1 for o in couts)
            print(f"co-runner {cn:36s} victim {vn:48s} [{kern[vn]:26s}]: " +
                  ("bit-identical (15 outputs)" if bad == 0 else f"{bad} of 15 outputs WRONG") + f"; co-runner wrong: {cobad} of 10", flush=True)
