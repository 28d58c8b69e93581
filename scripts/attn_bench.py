"""Attention microbench on the four call shapes of the headline workload (per gene, B=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variantformer_amd import ops, _lib
from variantformer_amd.seq2gene.modules.layers import get_alibi_slopes
if os.environ.get("VF_LIB"):                 # A/B against another build of the library on the same box
    _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["VF_LIB"]))
QLOG2 = os.environ.get("ATTN_QLOG2", "0") == "1"      # q already carries the softmax scale (the model's own call form)

def run(name, H, dh, ql, kl, alibi, self_attn, reps=20):
    D = H * dh
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32, device="cuda")
    slopes = get_alibi_slopes(H).float().cuda() if alibi else None
    if self_attn:
        qkv = (torch.randn((tq, 3 * D), device="cuda")).bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q = torch.randn((tq, D), device="cuda").bfloat16()
        kv = torch.randn((tk, 2 * D), device="cuda").bfloat16()
        k, v = kv[:, :D], kv[:, D:]
    for _ in range(3):
        ops.attn_varlen(q, k, v, cu_q, cu_k, max(ql), max(kl), H, dh, slopes, q_log2=QLOG2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.attn_varlen(q, k, v, cu_q, cu_k, max(ql), max(kl), H, dh, slopes, q_log2=QLOG2)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    flops = 4.0 * sum(a * b for a, b in zip(ql, kl)) * D
    nbytes = 2.0 * D * (2 * tq + 2 * tk)
    print("%-26s %8.1f us  %7.1f TFLOP/s  %7.1f GB/s (algorithmic)" % (name, us, flops / us / 1e6, nbytes / us / 1e3))

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(0)
run("gene->CRE cross (dh48)", 32, 48, [54 * 201] * g, [1024] * g, False, False)
run("gene self (dh48,alibi)", 32, 48, [201] * (54 * g), [201] * (54 * g), True, True)
run("CRE self (dh48,alibi)", 32, 48, [1024] * g, [1024] * g, True, True)
run("CRE ctx cross (dh48)", 32, 48, [1024] * g, [1024] * g, False, False)
cl = list(rng.integers(70, 126, 1024 * g))
run("seq2reg CRE windows (dh64)", 8, 64, cl, cl, False, True)
gl = [200] * (200 * g)
run("seq2reg gene chunks (dh64)", 8, 64, gl, gl, False, True)
