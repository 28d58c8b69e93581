import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops
for rows, D in ((86832, 1536), (769460, 512), (8192, 1536)):
    x = torch.randn((rows, D), device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    for _ in range(3): ops.layernorm(x, g, b)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.layernorm(x, g, b)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"LN rows={rows} D={D}: {us:8.1f} us  {rows * D * 6 / us / 1e3:7.0f} GB/s")
