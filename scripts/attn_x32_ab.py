"""gene -> CRE cross attention (dh 48, no bias, q pre-scaled: attn_x32_kernel<.., FAST>) at the bench's batch: timing, and
bit-identity of two library builds.  usage: python scripts/attn_x32_ab.py [genes]   (VF_LIB=<name>.so picks the build;
run once per build and compare the checksums, or VF_AB=libvf_hip_prev.so to alternate both builds in child processes)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VF_AB"):
    g = sys.argv[1] if len(sys.argv) > 1 else "32"
    for rep in range(2):
        for lib in (os.environ["VF_AB"], "libvf_hip.so"):
            env = dict(os.environ, VF_LIB=lib)
            env.pop("VF_AB")
            print(subprocess.run([sys.executable, __file__, g], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
import numpy as np, torch
from variantformer_amd import ops, _lib
name = os.environ.get("VF_LIB", "libvf_hip.so")
_lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), name))
g = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, dh = 32, 48
D = H * dh
for tag, ql, kl in (("gene->CRE cross", [54 * 201] * g, [1024] * g), ("ragged cross", [54 * 201 - 7 * i for i in range(g)], [1024 - 13 * i for i in range(g)])):
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(1)
    q = (torch.randn((tq, D), device="cuda", generator=gen) * 0.35).bfloat16()      # already carries scale * log2 e
    kv = torch.randn((tk, 2 * D), device="cuda", generator=gen).bfloat16()
    k, v = kv[:, :D], kv[:, D:]
    out = None
    for _ in range(3):
        out = ops.attn_varlen(q, k, v, cu_q, cu_k, max(ql), max(kl), H, dh, None, q_log2=True)
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            ops.attn_varlen(q, k, v, cu_q, cu_k, max(ql), max(kl), H, dh, None, q_log2=True)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 4 * 1e3)
    flops = 4.0 * sum(a * b for a, b in zip(ql, kl)) * D
    chk = int(out.view(torch.int16).to(torch.int64).sum())
    print("%-18s %-20s %8.1f us  %7.1f TFLOP/s  checksum %d" % (name, tag, best, flops / best / 1e6, chk))
