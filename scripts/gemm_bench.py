"""GEMM tile-configuration sweep on the shapes of the headline workload (random data, interleaved rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops, _lib
if os.environ.get("VF_TUNING_LIB"):          # diagnostic / sweep variants live in libvf_hip_tuning.so only
    from variantformer_amd.csrc.build import TUNING_LIB
    _lib.load(TUNING_LIB)

SHAPES = [  # (name, M, N, K, epilogue)
    ("gene Wqkv", 10854, 4608, 1536, ops.EPI_BF16), ("gene out_proj", 10854, 1536, 1536, ops.EPI_RES_F32),
    ("gene geglu1", 10854, 2048, 1536, ops.EPI_GEGLU_BF16), ("gene geglu2", 10854, 1536, 1024, ops.EPI_RES_F32),
    ("s2r Wqkv", 140000, 1536, 512, ops.EPI_BF16), ("s2r out_proj", 140000, 512, 512, ops.EPI_RES_F32),
    ("s2r geglu1", 140000, 2048, 512, ops.EPI_GEGLU_BF16), ("s2r geglu2", 140000, 512, 1024, ops.EPI_RES_F32),
    ("square 8k", 8192, 8192, 8192, ops.EPI_BF16),
    ("s2r8 Wqkv", 769460, 1536, 512, ops.EPI_BF16), ("s2r8 geglu", 769460, 2048, 512, ops.EPI_GEGLU_BF16),
    ("gene8 Wqkv", 86832, 4608, 1536, ops.EPI_BF16), ("gene8 out_proj", 86832, 1536, 1536, ops.EPI_RES_F32),
    ("gene8 geglu", 86832, 2048, 1536, ops.EPI_GEGLU_BF16), ("gene8 ff", 86832, 1536, 1024, ops.EPI_RES_F32),
    ("gene8 Wq ", 86832, 1536, 1536, ops.EPI_BF16),
    ("s2r8 out_proj", 769460, 512, 512, ops.EPI_RES_F32), ("s2r8 ff", 769460, 512, 1024, ops.EPI_RES_F32),
    ("cre8 kv", 8192, 3072, 1536, ops.EPI_BF16), ("cre8 Wq", 8192, 1536, 1536, ops.EPI_BF16),
    ("cre8 geglu", 8192, 2048, 1536, ops.EPI_GEGLU_BF16), ("cre8 ff", 8192, 1536, 1024, ops.EPI_RES_F32),
    ("cre Wqkv", 1024, 4608, 1536, ops.EPI_BF16), ("cre out_proj", 1024, 1536, 1536, ops.EPI_RES_F32),
    ("cre kv", 1024, 3072, 1536, ops.EPI_BF16), ("cre geglu2", 1024, 1536, 1024, ops.EPI_RES_F32),
    ("cre8 Wqkv", 8192, 4608, 1536, ops.EPI_BF16), ("cre8 out_proj", 8192, 1536, 1536, ops.EPI_RES_F32),
]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,20".split(","))]
rounds = 5
only = sys.argv[2] if len(sys.argv) > 2 else None
print("%-14s %7s %5s %5s | " % ("shape", "M", "N", "K") + " ".join("v%d TF/s" % v for v in variants))
for name, M, N, K, epi in SHAPES:
    if only and only not in name: continue
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    res = torch.rand((M, N), device="cuda") if epi == ops.EPI_RES_F32 else None
    ref = None
    best = {v: 1e9 for v in variants}
    for rnd in range(rounds + 1):
        for v in variants:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            o = ops.gemm(a, w, b, epi, residual=res, variant=v)
            e.record()
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = o.float()
                else:
                    err = float((o.float() - ref).abs().max() / ref.abs().max())
                    assert err < 1e-2 or v > 100, (name, v, err)
            else:
                best[v] = min(best[v], s.elapsed_time(e))
    fl = 2.0 * M * N * K
    print("%-14s %7d %5d %5d | " % (name, M, N, K) + " ".join("%8.0f" % (fl / (best[v] * 1e-3) / 1e12) for v in variants))
