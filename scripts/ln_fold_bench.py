"""LayerNorm-fold micro-benchmark: the folded producer / consumer GEMMs against the plain GEMM (+ the separate LayerNorm
pass they replace) on the shapes of the headline workload.  Random data, interleaved rounds, best of 5."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops, _lib
if os.environ.get("VF_TUNING_LIB"):          # epilogue probes (VF_G8_DBG) live in libvf_hip_tuning.so only
    from variantformer_amd.csrc.build import TUNING_LIB
    _lib.load(TUNING_LIB)
if os.environ.get("VF_LIB"):                 # A/B against another build of the library on the same box
    _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["VF_LIB"]))

PRODUCERS = [("gene8 out_proj", 86832, 1536, 1536), ("gene8 ff", 86832, 1536, 1024), ("s2r8 out_proj", 769460, 512, 512),
             ("s2r8 ff", 769460, 512, 1024), ("cre8 out_proj", 8192, 1536, 1536), ("cre8 ff", 8192, 1536, 1024)]
CONSUMERS = [("gene8 Wqkv", 86832, 4608, 1536, False), ("gene8 Wq", 86832, 1536, 1536, False),
             ("gene8 geglu", 86832, 2048, 1536, True), ("s2r8 Wqkv", 769460, 1536, 512, False),
             ("s2r8 geglu", 769460, 2048, 512, True), ("cre8 Wqkv", 8192, 4608, 1536, False)]


def best_of(fns, rounds=5):
    best = [1e9] * len(fns)
    for r in range(rounds + 1):
        for i, f in enumerate(fns):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            f()
            e.record()
            torch.cuda.synchronize()
            if r:
                best[i] = min(best[i], s.elapsed_time(e))
    return best


print("producer        M     N    K | plain us  folded us (gemm+finalize)  layernorm us | plain TF  folded TF")
for name, M, N, K in PRODUCERS:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    res = torch.rand((M, N), device="cuda")
    g, be = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    x = torch.rand((M, N), device="cuda")
    t = best_of([lambda: ops.gemm(a, w, b, ops.EPI_RES_F32, residual=res), lambda: ops.gemm_ln_producer(a, w, b, res),
                 lambda: ops.layernorm(x, g, be)])
    fl = 2.0 * M * N * K
    print("%-14s %6d %5d %4d | %8.0f %10.0f %26.0f | %8.0f %9.0f" % (name, M, N, K, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3,
                                                                     fl / t[0] / 1e9, fl / t[1] / 1e9))
print("consumer        M     N    K | plain us  folded us | plain TF  folded TF")
for name, M, N, K, geglu in CONSUMERS:
    x = torch.rand((M, K), device="cuda") * 2 - 1
    s = ops.ln_stream(x)
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    cs = w.float().sum(dim=1).contiguous()
    epi = ops.EPI_GEGLU_BF16 if geglu else ops.EPI_BF16
    t = best_of([lambda: ops.gemm(s.x16, w, b, epi), lambda: ops.gemm_ln_consumer(s, w, b, cs, epi)])
    fl = 2.0 * M * N * K
    print("%-14s %6d %5d %4d | %8.0f %9.0f | %8.0f %9.0f" % (name, M, N, K, t[0] * 1e3, t[1] * 1e3, fl / t[0] / 1e9, fl / t[1] / 1e9))
