import sys, os
sys.path.insert(0, os.getcwd())
import torch
from variantformer_amd import ops
M, N = 769460, 1536
for K in (64, 128, 256, 512, 1024):
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    for epi, name in ((ops.EPI_BF16, "bf16"), (ops.EPI_F32, "f32")):
        best = 1e9
        for r in range(6):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.gemm(a, w, b, epi, variant=20); e.record(); torch.cuda.synchronize()
            if r: best = min(best, s.elapsed_time(e))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"K={K:5d} {name}: {best*1e3:8.1f} us  per tile-round {best*1e3/(tiles/256):6.2f} us")
