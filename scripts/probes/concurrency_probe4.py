"""Round 6, concurrency probe 4: why does the LayerNorm-consumer GEMM of gemm_mfma_kernel go wrong beside a co-resident
attention kernel (probe 3)?  Run once per library build of scripts/probes/build_probe_libs.py:
    python scripts/probes/concurrency_probe4.py [path/to/libvf_*.so]
Victims (side stream, 3 launches a round, 5 rounds): LayerNorm consumer 16-bit out (128x128 tiles), N = 320 fp32 (64x64), plain
GEMM 128x128.  Co-runners (main stream): cross attention on the 32x32x16 kernel, the same on the tiled 16x16x32 kernel
(VF_ATTN_X32=0), a plain 128x128 GEMM, a LayerNorm-consumer GEMM.  The co-runner's own output is checked too, and the first
wrong victim output is dissected: which (row block, column) units, and what the wrong values equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from variantformer_amd import _lib
libpath = next((a for a in sys.argv[1:] if a.endswith(".so")), None)
if libpath:
    _lib.load(os.path.abspath(libpath))
from variantformer_amd import ops

print(f"library: {libpath or _lib.LIB_PATH}", flush=True)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
def rnd(*shape, scale=1.0):
    return (torch.rand(shape, device=dev, generator=g) * 2 - 1) * scale
G = 8
M, N, K = 1024 * G, 4608, 1536
a, w, b, c = rnd(M, K).bfloat16(), (rnd(N, K) / K ** 0.5).bfloat16(), rnd(N), rnd(N)
s = ops.ln_stream(rnd(M, K))
w320, b320, c320 = (rnd(320, K) / K ** 0.5).bfloat16(), rnd(320), rnd(320)
victims = {
    "LayerNorm consumer, 16-bit out": lambda: ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16),
    "LayerNorm consumer, N = 320 fp32 out": lambda: ops.gemm_ln_consumer(s, w320, b320, c320, ops.EPI_F32),
    "plain GEMM, 128x128 (variant 1)": lambda: ops.gemm(a, w, b, ops.EPI_BF16, variant=1),
}
H, dh = 32, 48
D = H * dh
ql, kl = [54 * 201] * G, [1024] * G
cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32, device=dev)
cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32, device=dev)
q, kv = rnd(sum(ql), D, scale=0.35).bfloat16(), rnd(sum(kl), 2 * D).bfloat16()
a2, w2, b2 = rnd(54 * 201 * G, K).bfloat16(), (rnd(1536, K) / K ** 0.5).bfloat16(), rnd(1536)
s2 = ops.ln_stream(rnd(54 * 201 * G, K))
c2 = rnd(1536)


def attn(x32):
    os.environ["VF_ATTN_X32"] = "1" if x32 else "0"
    return ops.attn_varlen(q, kv[:, :D], kv[:, D:], cu_q, cu_k, max(ql), max(kl), H, dh, None, q_log2=True)


corunners = {
    "cross attention, 32x32x16 kernel": lambda: attn(True),
    "cross attention, tiled 16x16x32 kernel": lambda: attn(False),
    "plain GEMM 128x128 (variant 1)": lambda: ops.gemm(a2, w2, b2, ops.EPI_BF16, variant=1),
    "LayerNorm consumer GEMM 128x128": lambda: ops.gemm_ln_consumer(s2, w2, b2, c2, ops.EPI_BF16),
}


def dissect(o, ref, vn):
    bad = (o != ref).nonzero()
    rows, cols = bad[:, 0], bad[:, 1]
    units = {}
    for r_, c_ in zip(rows.tolist(), cols.tolist()):
        units.setdefault((r_ // 16, c_), []).append(r_)
    full = sum(1 for v in units.values() if len(v) == 16)
    print(f"    {len(bad)} wrong elements in {len(units)} (16-row block, column) units, {full} of them all 16 rows; "
          f"columns mod 16: {sorted(set(c_ % 16 for _, c_ in units))}; distinct 128-row tiles: {len(set(rb // 8 for rb, _ in units))}")
    if "16-bit out" not in vn:
        return
    # candidates for a wrong value of out[m][n] = acc*rstd + (mu*colsum[n] + bias[n]),  mu = -mean*rstd
    st = s.stats.float()
    mu, rs = -st[:, 0] * st[:, 1], st[:, 1]
    for (rb, n), rr in list(units.items())[:6]:
        m = torch.tensor(rr, device=dev)
        got, want = o[m, n].float(), ref[m, n].float()
        acc = (want - (mu[m] * c[n] + b[n])) / rs[m]                 # accumulator, to 16-bit rounding of `want`
        cands = {"no bias": acc * rs[m] + mu[m] * c[n], "no colsum term": acc * rs[m] + b[n], "acc*rstd only": acc * rs[m],
                 "acc + bias (plain epilogue)": acc + b[n], "t only": mu[m] * c[n] + b[n], "acc only": acc}
        for dn in (-3, -2, -1, 1, 2, 3):
            if 0 <= n + dn < N:
                cands[f"bias, colsum of column {dn:+d}"] = acc * rs[m] + (mu[m] * c[n + dn] + b[n + dn])
                cands[f"bias of column {dn:+d}"] = acc * rs[m] + (mu[m] * c[n] + b[n + dn])
                cands[f"colsum of column {dn:+d}"] = acc * rs[m] + (mu[m] * c[n + dn] + b[n])
        best = min(cands.items(), key=lambda kv_: float((kv_[1] - got).abs().max()))
        print(f"      rows {rr[0]}..{rr[-1]} col {n}: got-want = {[round(float(x), 3) for x in (got - want)[:6]]}..  "
              f"closest: '{best[0]}' (max |d| {float((best[1] - got).abs().max()):.3g}); bias {float(b[n]):.3f} colsum {float(c[n]):.3f}")


with torch.no_grad():
    refs = {n: f().clone() for n, f in victims.items()}
    corefs = {n: f().clone() for n, f in corunners.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for cn, co in corunners.items():
        for vn, vf in victims.items():
            bad = worst = cobad = 0
            first = None
            for rep in range(5):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    outs = [vf() for _ in range(3)]
                couts = [co() for _ in range(2)]
                main.wait_stream(side)
                torch.cuda.synchronize()
                for o in outs:
                    if not torch.equal(o, refs[vn]):
                        bad += 1
                        worst = max(worst, int((o != refs[vn]).sum()))
                        first = o if first is None else first
                cobad += sum(0 if torch.equal(o, corefs[cn]) else 1 for o in couts)
            print(f"co-runner {cn:40s} victim {vn:38s}: " +
                  ("bit-identical (15 outputs)" if bad == 0 else f"{bad} of 15 outputs WRONG (up to {worst} elements)") +
                  f"; co-runner outputs wrong: {cobad} of 10", flush=True)
            if first is not None:
                dissect(first, refs[vn], vn)
