"""Launch one GEMM shape/variant a few times (target for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops
M, N, K, v = [int(x) for x in sys.argv[1:5]]
epi = int(sys.argv[5]) if len(sys.argv) > 5 else ops.EPI_BF16
a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
b = torch.rand((N,), device="cuda")
res = torch.rand((M, N), device="cuda") if epi == ops.EPI_RES_F32 else None
for _ in range(5):
    ops.gemm(a, w, b, epi, residual=res, variant=v)
torch.cuda.synchronize()
