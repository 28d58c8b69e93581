"""X-stationary K = 512 GEMM (variant 30) against the persistent 256x256 kernel on the seq2reg consumer shapes:
correctness against the other kernel on the same operands, then interleaved timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops

SHAPES = [("s2r8 Wqkv", 769460, 1536, 512, ops.EPI_BF16), ("s2r8 geglu", 769460, 2048, 512, ops.EPI_GEGLU_BF16),
          ("s2r8g Wqkv", 319459, 1536, 512, ops.EPI_BF16), ("ragged", 70001, 1536, 512, ops.EPI_BF16)]
for name, M, N, K, epi in SHAPES:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    ref = ops.gemm(a, w, b, epi, variant=22)
    got = ops.gemm(a, w, b, epi, variant=30)
    torch.cuda.synchronize()
    bad = int((ref != got).sum())
    err = float((ref.float() - got.float()).abs().max())
    best = {22: 1e9, 30: 1e9}
    for r in range(6):
        for v in (22, 30):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.gemm(a, w, b, epi, variant=v)
            e.record()
            torch.cuda.synchronize()
            if r:
                best[v] = min(best[v], s.elapsed_time(e))
    fl = 2.0 * M * N * K
    print("%-12s %7d %5d | v22 %8.1f us %6.0f TF | xs %8.1f us %6.0f TF | differing elements %d (max abs %.3g)" % (
        name, M, N, best[22] * 1e3, fl / best[22] / 1e9, best[30] * 1e3, fl / best[30] / 1e9, bad, err))
