"""Round 6, concurrency probe 3: which GEMM kernels go wrong beside which co-runner?  Victims (side stream, 3 launches each
round): the CRE-stream Wqkv shape as plain GEMM through each tile configuration (variants 1 = 128x128, 20 = one-shot 256x256,
22 = persistent 256x256), as LayerNorm consumer (auto), and the N = 320 fp32 logits consumer.  Co-runners (main stream): the
gene->CRE cross attention (attn_x32_kernel), a big persistent GEMM, and a pure memory hog (torch elementwise over 2 GB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from variantformer_amd import ops
from variantformer_amd.seq2gene.modules.layers import get_alibi_slopes

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
def rnd(*shape, scale=1.0):
    return (torch.rand(shape, device=dev, generator=g) * 2 - 1) * scale
G = 8
M, N, K = 1024 * G, 4608, 1536
a, w, b, c = rnd(M, K).bfloat16(), (rnd(N, K) / K ** 0.5).bfloat16(), rnd(N), rnd(N)
s = ops.ln_stream(rnd(M, K))
w320, b320, c320 = (rnd(320, K) / K ** 0.5).bfloat16(), rnd(320), rnd(320)
victims = {
    "plain GEMM, 128x128 (variant 1)": lambda: ops.gemm(a, w, b, ops.EPI_BF16, variant=1),
    "plain GEMM, one-shot 256x256 (variant 20)": lambda: ops.gemm(a, w, b, ops.EPI_BF16, variant=20),
    "plain GEMM, persistent 256x256 (variant 22)": lambda: ops.gemm(a, w, b, ops.EPI_BF16, variant=22),
    "plain GEMM fp32 out, one-shot (variant 20)": lambda: ops.gemm(a, w, b, ops.EPI_F32, variant=20),
    "LayerNorm consumer, 16-bit out (auto)": lambda: ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16),
    "LayerNorm consumer, GeGLU (auto)": lambda: ops.gemm_ln_consumer(s, w, b, c, ops.EPI_GEGLU_BF16),
    "LayerNorm consumer, N = 320 fp32 out": lambda: ops.gemm_ln_consumer(s, w320, b320, c320, ops.EPI_F32),
}
H, dh = 32, 48
D = H * dh
ql, kl = [54 * 201] * G, [1024] * G
cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32, device=dev)
cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32, device=dev)
q, kv = rnd(sum(ql), D, scale=0.35).bfloat16(), rnd(sum(kl), 2 * D).bfloat16()
big_a, big_w, big_b = rnd(54 * 201 * G, K).bfloat16(), (rnd(4608, K) / K ** 0.5).bfloat16(), rnd(4608)
hog = torch.zeros(512 * 1024 * 1024, dtype=torch.float32, device=dev)      # 2 GiB
corunners = {
    "gene->CRE cross attention (attn_x32)": lambda: ops.attn_varlen(q, kv[:, :D], kv[:, D:], cu_q, cu_k, max(ql), max(kl), H, dh, None, q_log2=True),
    "big persistent GEMM (gene Wqkv)": lambda: ops.gemm(big_a, big_w, big_b, ops.EPI_BF16, variant=22),
    "memory hog (torch add_ over 2 GiB)": lambda: hog.add_(1.0),
    "nothing (two streams, no co-runner)": lambda: None,
}
with torch.no_grad():
    refs = {n: f().clone() for n, f in victims.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for cn, co in corunners.items():
        for vn, vf in victims.items():
            bad = worst = 0
            for rep in range(5):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    outs = [vf() for _ in range(3)]
                for _ in range(2):
                    co()
                main.wait_stream(side)
                torch.cuda.synchronize()
                for o in outs:
                    if not torch.equal(o, refs[vn]):
                        bad += 1
                        worst = max(worst, int((o != refs[vn]).sum()))
            print(f"co-runner {cn:40s} victim {vn:46s} [{ops.last_kernel('gemm'):26s}]: " +
                  ("bit-identical (15 outputs)" if bad == 0 else f"{bad} of 15 outputs WRONG (up to {worst} elements)"), flush=True)
