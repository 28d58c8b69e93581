"""gemm4_kernel (two 4-wave blocks per CU, variant 40) against the 8-wave 256x256 kernels (variants 20 / 22):
bit-identity on ragged shapes, an exact-integer race screen, and interleaved timing on the workload's shapes.
usage: python scripts/gemm4_probe.py [check|time|all] [name filter]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops, _lib
if os.environ.get("VF_TUNING_LIB"):
    from variantformer_amd.csrc.build import TUNING_LIB
    _lib.load(TUNING_LIB)

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
only = sys.argv[2] if len(sys.argv) > 2 else None
dev = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) * 2 - 1) * scale)


def check():
    bad = 0
    cases = [(515, 776, 256), (700, 520, 128), (256 * 5 + 3, 128 * 7, 512), (4099, 1536, 1536), (130, 136, 384),
             (256 * 40 + 19, 2048, 512), (77, 40, 128)]
    for (M, N, K) in cases:
        a = rnd((M, K), 1).cuda().bfloat16()
        w = rnd((N, K), 2, 1 / math.sqrt(K)).cuda().bfloat16()
        b = rnd((N,), 3, 0.5).cuda()
        res = rnd((M, N), 4).cuda()
        for epi, name in [(ops.EPI_BF16, "bf16"), (ops.EPI_F32, "f32"), (ops.EPI_RES_F32, "res"), (ops.EPI_GELU_F32, "gelu")]:
            o22 = ops.gemm(a, w, b, epi, residual=res if epi == ops.EPI_RES_F32 else None, variant=20)
            o40 = ops.gemm(a, w, b, epi, residual=res if epi == ops.EPI_RES_F32 else None, variant=40)
            torch.cuda.synchronize()
            same = torch.equal(o22, o40)
            ref = a.float() @ w.float().t() + b
            if epi == ops.EPI_RES_F32: ref = ref + res
            if epi == ops.EPI_GELU_F32: ref = torch.nn.functional.gelu(ref)
            err = float((o40.float() - ref).abs().max())
            print(f"check M={M} N={N} K={K} {name}: identical to v20 {same}, max abs err vs fp32 {err:.3e}")
            bad += (not same)
        if N % 32 == 0:
            wp, bp = ops.pack_geglu_rows(w, b)
            o22 = ops.gemm(a, wp, bp, ops.EPI_GEGLU_BF16, variant=20)
            o40 = ops.gemm(a, wp, bp, ops.EPI_GEGLU_BF16, variant=40)
            torch.cuda.synchronize()
            same = torch.equal(o22, o40)
            print(f"check M={M} N={N} K={K} geglu: identical to v20 {same}")
            bad += (not same)
    # exact-integer race screen over several waves of tiles, repeated
    M, N, K = 256 * 75 + 19, 128 * 18, 1536
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    ab, wb = a.cuda().bfloat16(), w.cuda().bfloat16()
    ref = a.cuda() @ w.cuda().t()
    for K2 in (1536, 512, 128):
        refk = a.cuda()[:, :K2] @ w.cuda()[:, :K2].t()
        abk, wbk = ab[:, :K2].contiguous(), wb[:, :K2].contiguous()
        for it in range(6):
            out = ops.gemm(abk, wbk, None, ops.EPI_F32, variant=40)
            torch.cuda.synchronize()
            nbad = int((out != refk).sum())
            if nbad:
                print(f"RACE SCREEN K={K2} iteration {it}: {nbad} wrong elements")
                bad += 1
        print(f"race screen K={K2}: done")
    print("CHECK", "FAILED" if bad else "ok")
    return bad


SHAPES = [  # (name, M, N, K, epilogue) at 32 genes per step
    ("gene Wqkv", 347328, 4608, 1536, ops.EPI_BF16), ("gene Wq", 347328, 1536, 1536, ops.EPI_BF16),
    ("gene geglu", 347328, 2048, 1536, ops.EPI_GEGLU_BF16), ("gene out_proj", 347328, 1536, 1536, ops.EPI_RES_F32),
    ("gene ff", 347328, 1536, 1024, ops.EPI_RES_F32),
    ("s2r Wqkv", 3080279, 1536, 512, ops.EPI_BF16), ("s2r geglu", 3080279, 2048, 512, ops.EPI_GEGLU_BF16),
    ("s2r out_proj", 3080279, 512, 512, ops.EPI_RES_F32), ("s2r ff", 3080279, 512, 1024, ops.EPI_RES_F32),
    ("cre Wqkv", 32768, 4608, 1536, ops.EPI_BF16), ("square 8k", 8192, 8192, 8192, ops.EPI_BF16),
]


def timeit():
    variants = [int(v) for v in os.environ.get("VARIANTS", "22,40,20").split(",")]
    rounds = 5
    print("%-14s %8s %5s %5s | " % ("shape", "M", "N", "K") + " ".join("v%d TF/s" % v for v in variants))
    for name, M, N, K, epi in SHAPES:
        if only and only not in name: continue
        a = (torch.rand((M, K), device=dev) * 2 - 1).bfloat16()
        w = ((torch.rand((N, K), device=dev) * 2 - 1) / K ** 0.5).bfloat16()
        b = torch.rand((N,), device=dev)
        res = torch.rand((M, N), device=dev) if epi == ops.EPI_RES_F32 else None
        out = torch.empty((M, N // 2 if epi == ops.EPI_GEGLU_BF16 else N), device=dev,
                          dtype=torch.float32 if epi == ops.EPI_RES_F32 else torch.bfloat16)
        best = {v: 1e9 for v in variants}
        for rd in range(rounds + 1):
            for v in variants:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.gemm(a, w, b, epi, residual=res, out=out, variant=v)
                e.record()
                torch.cuda.synchronize()
                if rd: best[v] = min(best[v], s.elapsed_time(e))
        fl = 2.0 * M * N * K
        print("%-14s %8d %5d %5d | " % (name, M, N, K) + " ".join("%8.0f" % (fl / (best[v] * 1e-3) / 1e12) for v in variants), flush=True)
        del a, w, b, res, out
        torch.cuda.empty_cache()


if mode in ("check", "all"):
    check()
if mode in ("time", "all"):
    timeit()
