import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops
H, dh = 32, 48; D = H*dh
tq, tk = 54*201, 1024
q = torch.randn((tq, D), device="cuda").bfloat16(); kv = torch.randn((tk, 2*D), device="cuda").bfloat16()
cu_q = torch.tensor([0, tq], dtype=torch.int32, device="cuda"); cu_k = torch.tensor([0, tk], dtype=torch.int32, device="cuda")
for _ in range(5): ops.attn_varlen(q, kv[:, :D], kv[:, D:], cu_q, cu_k, tq, tk, H, dh)
torch.cuda.synchronize()
