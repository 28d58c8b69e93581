"""Timing of the LayerNorm-producer GEMMs (attention out-projections with a 16-bit residual and no fp32 store; the fp32
trunk's down-projections) on the bench's shapes, for A/B runs of the tuning library's start-up stagger:
  VF_LIB=libvf_hip_tuning.so VF_G8_STAGGER=0|3|6|9 python scripts/producer_stagger_bench.py
(every other CU of an XCD starts its first tile VF_G8_STAGGER x ~4 us late; profiles/r03_*stagger*.log)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops, _lib
if os.environ.get("VF_LIB"):
    _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["VF_LIB"]))

SHAPES = [("gene8 out_proj r16-nox", 86832, 1536, 1536, "r16"), ("gene8 down (fp32 trunk)", 86832, 1536, 1024, "f32"),
          ("s2r8 out_proj r16-nox", 769460, 512, 512, "r16"), ("s2r8 down (fp32 trunk)", 769460, 512, 1024, "f32")]
for name, M, N, K, kind in SHAPES:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    res = torch.rand((M, N), device="cuda")
    stream = ops.ln_stream(res)
    best = 1e9
    for r in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if kind == "r16":
            ops.gemm_ln_producer(a, w, b, stream, need_x=False)
        else:
            ops.gemm_ln_producer(a, w, b, res)
        e.record()
        torch.cuda.synchronize()
        if r:
            best = min(best, s.elapsed_time(e))
    print("%-26s %7d %5d %5d  %8.1f us  %7.0f TFLOP/s   [stagger=%s]" % (
        name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9, os.environ.get("VF_G8_STAGGER", "0")))
