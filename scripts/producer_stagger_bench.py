"""Timing of the fp32-residual LayerNorm producers (trunk down-projection etc.) on the bench's shapes.
Round 3 used it for an A/B that is NOT in the tree any more: the persistent 256x256 kernel with a persistent phase offset
(0 / 6 / 12 / 18 / 26 us) between the two halves of every XCD's CUs, so that one half's epilogue would run under the other
half's main loop.  Result (profiles/r03_d_producer_stagger_experiment.log): +-1 % on every shape -- the epilogue is bound
by the bytes ONE CU keeps in flight (~25 GB/s per CU whatever the other CUs do), not by all CUs hitting HBM together.
  VF_GEMM_PERSIST=1|2 python scripts/producer_stagger_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops

SHAPES = [("gene8 down", 86832, 1536, 1024), ("gene8 out_proj(fp32 res)", 86832, 1536, 1536), ("s2r8 down", 769460, 512, 1024),
          ("s2r8 out_proj", 769460, 512, 512)]
for name, M, N, K in SHAPES:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    res = torch.rand((M, N), device="cuda")
    best = 1e9
    for r in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gemm_ln_producer(a, w, b, res)
        e.record()
        torch.cuda.synchronize()
        if r:
            best = min(best, s.elapsed_time(e))
    print("%-26s %7d %5d %5d  %8.1f us  %7.0f TFLOP/s   [persist=%s stagger=%s]" % (
        name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9, os.environ.get("VF_GEMM_PERSIST", "1"),
        os.environ.get("VF_G8X_STAGGER", "0")))
