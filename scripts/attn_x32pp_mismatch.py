import sys, os, math
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from variantformer_amd import ops
torch.manual_seed(0)
dh, H = 48, 32; D = H*dh
c = math.log2(math.e)/math.sqrt(dh)
ql = [3000, 33, 2049, 700, 1, 2500, 600, 300, 256]; kl = [1024, 70, 63, 64, 200, 65, 0, 31, 1000]
cu_q = torch.tensor([0]+list(np.cumsum(ql)), dtype=torch.int32); cu_k = torch.tensor([0]+list(np.cumsum(kl)), dtype=torch.int32)
q = (torch.randn(sum(ql), D)*2*c).bfloat16().cuda(); kv = (torch.randn(sum(kl), 2*D)*2).bfloat16().cuda()
for extreme in (False, True):
    qq = q.clone()
    if extreme: qq[100] *= 15.0
    out = ops.attn_varlen(qq, kv[:, :D], kv[:, D:], cu_q.cuda(), cu_k.cuda(), max(ql), max(kl), H, dh, q_log2=True)
    n4, k4 = int(cu_q[4]), int(cu_k[4])
    sub = ops.attn_varlen(qq[:n4], kv[:k4, :D], kv[:k4, D:], cu_q[:5].cuda(), cu_k[:5].cuda(), max(ql[:4]), max(kl[:4]), H, dh, q_log2=True)
    torch.cuda.synchronize()
    ne = (out[:n4].view(torch.int16) != sub.view(torch.int16)).any(dim=1).cpu().numpy()
    rows = np.nonzero(ne)[0]
    print("extreme row" if extreme else "ordinary", ": rows differing", len(rows), "first/last", rows[:5], rows[-5:] if len(rows) else "", "max abs diff", float((out[:n4].float()-sub.float()).abs().max()))
