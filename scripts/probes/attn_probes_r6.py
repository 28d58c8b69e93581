"""Round-6 attention probes on the bench's shapes (32 genes), tuning library, interleaved rounds in one process:
  (1) seq2reg self attention (dh 64, <= 125-token windows / 200-token chunks): the product kernel against its own LOAD / STORE
      skeleton (VF_ATTN_SHORT_DBG=1: Q / K / V loads, LDS staging, output stores, no tile arithmetic) at 3, 2 and 1 resident
      blocks per CU (VF_ATTN_SHORT_LDS pads the LDS request) -- the streaming ceiling of the one-block-per-(window, head)
      structure and how it scales with the bytes in flight;
  (2) gene self attention (dh 48, ALiBi, 201 tokens): the same;
  (3) gene -> CRE cross attention (dh 48, no bias): v_mfma_f32_32x32x16 kernel (attn_x32_kernel) against the 16x16x32 kernel
      (attn_fwd_kernel, VF_ATTN_X32=0) -- the round-5 verdict's "other MFMA shape" on this body.
Results of the DBG launches are meaningless."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from variantformer_amd import ops, _lib
from variantformer_amd.csrc.build import TUNING_LIB
from variantformer_amd.seq2gene.modules.layers import get_alibi_slopes
_lib.load(TUNING_LIB)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)


def case(name, H, dh, ql, kl, alibi, self_attn, modes):
    D = H * dh
    tq, tk = sum(ql), sum(kl)
    cu_q = torch.tensor([0] + list(np.cumsum(ql)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(kl)), dtype=torch.int32, device="cuda")
    slopes = get_alibi_slopes(H).float().cuda() if alibi else None
    if self_attn:
        qkv = (torch.randn((tq, 3 * D), device="cuda") * 0.5).bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q = (torch.randn((tq, D), device="cuda") * 0.35).bfloat16()
        kv = torch.randn((tk, 2 * D), device="cuda").bfloat16()
        k, v = kv[:, :D], kv[:, D:]
    times = {m: [] for m in modes}
    kern = {}
    for rnd in range(6):
        for m, env in modes.items():
            for kk in ("VF_ATTN_SHORT_DBG", "VF_ATTN_SHORT_LDS", "VF_ATTN_X32"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                ops.attn_varlen(q, k, v, cu_q, cu_k, max(ql), max(kl), H, dh, slopes, q_log2=True)
            e.record()
            torch.cuda.synchronize()
            kern[m] = ops.last_kernel("attn")
            if rnd:
                times[m].append(s.elapsed_time(e) / 3)
    flops = 4.0 * sum(a * b for a, b in zip(ql, kl)) * D
    nbytes = 2.0 * D * (2 * tq + 2 * tk)
    for m in modes:
        t = sorted(times[m])
        print(f"{name:34s} {m:34s} min {t[0] * 1e3:8.1f} us  median {t[len(t) // 2] * 1e3:8.1f} us  {flops / t[0] / 1e9:7.1f} TFLOP/s  "
              f"{nbytes / t[0] / 1e9:6.2f} TB/s (algorithmic)  [{kern[m]}]", flush=True)


occ = {"product (3 blocks / CU)": {}, "load/store skeleton, 3 blocks / CU": {"VF_ATTN_SHORT_DBG": "1"},
       "product, 2 blocks / CU": {"VF_ATTN_SHORT_LDS": "70000"}, "skeleton, 2 blocks / CU": {"VF_ATTN_SHORT_DBG": "1", "VF_ATTN_SHORT_LDS": "70000"},
       "product, 1 block / CU": {"VF_ATTN_SHORT_LDS": "100000"}, "skeleton, 1 block / CU": {"VF_ATTN_SHORT_DBG": "1", "VF_ATTN_SHORT_LDS": "100000"}}
cl = [int(x) for x in rng.integers(70, 126, 1024 * G)]
case("seq2reg CRE windows (dh 64)", 8, 64, cl, cl, False, True, occ)
gl = [200] * (200 * G)
case("seq2reg gene chunks (dh 64)", 8, 64, gl, gl, False, True,
     {"product (2 blocks / CU)": {}, "load/store skeleton, 2 blocks / CU": {"VF_ATTN_SHORT_DBG": "1"},
      "product, 1 block / CU": {"VF_ATTN_SHORT_LDS": "100000"}, "skeleton, 1 block / CU": {"VF_ATTN_SHORT_DBG": "1", "VF_ATTN_SHORT_LDS": "100000"}})
case("gene self (dh 48, ALiBi, 201)", 32, 48, [201] * (54 * G), [201] * (54 * G), True, True, occ)
case("gene -> CRE cross (dh 48)", 32, 48, [54 * 201] * G, [1024] * G, False, False,
     {"v_mfma 32x32x16 (attn_x32_kernel)": {}, "v_mfma 16x16x32 (attn_fwd_kernel)": {"VF_ATTN_X32": "0"}})
