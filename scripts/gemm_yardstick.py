"""Yardstick only (never shipped): vendor bf16 GEMM via torch on the workload's shapes vs vf_gemm_bf16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops
SH = [("gene Wqkv", 10854, 4608, 1536), ("gene out_proj", 10854, 1536, 1536), ("gene geglu1", 10854, 2048, 1536),
      ("gene geglu2", 10854, 1536, 1024), ("s2r Wqkv", 140000, 1536, 512), ("s2r out_proj", 140000, 512, 512),
      ("s2r geglu1", 140000, 2048, 512), ("s2r geglu2", 140000, 512, 1024), ("cre8 Wqkv", 8192, 4608, 1536),
      ("square 8k", 8192, 8192, 8192), ("square 4k", 4096, 4096, 4096),
      ("gene8 Wqkv", 86832, 4608, 1536), ("gene8 Wq", 86832, 1536, 1536), ("gene8 geglu1", 86832, 2048, 1536),
      ("s2r8 Wqkv", 769460, 1536, 512), ("s2r8 geglu1", 769460, 2048, 512)]
def t(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, M, N, K in SH:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    ms_v = t(lambda: torch.matmul(a, w.t()))
    ms_m = t(lambda: ops.gemm(a, w, None, ops.EPI_BF16))
    fl = 2.0 * M * N * K
    print("%-14s %7d %5d %5d | vendor %6.0f TF/s | vf_gemm_bf16 %6.0f TF/s" % (name, M, N, K, fl / ms_v / 1e9, fl / ms_m / 1e9))
