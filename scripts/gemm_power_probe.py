"""Is the 256 x 256 GEMM held back by its instruction stream or by the clock the chip holds under it?  The same launch on random
operands, on all-zero operands and on a constant (identical code, identical memory traffic, different switching activity in the
matrix pipe), each sustained for >= 2 s so that the power management settles (MI355X_MICROARCH.md, DVFS give-back items 1 and 6)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import ops
SHAPES = [("gene Wqkv (32 genes)", 347328, 4608, 1536), ("seq2reg Wqkv (32 genes)", 3080279, 1536, 512), ("square 8k", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    row = []
    for kind in ("random", "zeros", "ones"):
        if kind == "random":
            a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
            w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
        else:
            v = 0.0 if kind == "zeros" else 1.0
            a = torch.full((M, K), v, device="cuda", dtype=torch.bfloat16)
            w = torch.full((N, K), v / K if v else 0.0, device="cuda", dtype=torch.bfloat16)
        b = torch.zeros((N,), device="cuda")
        out = ops.gemm(a, w, b, ops.EPI_BF16)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 2.0
        while time.perf_counter() < t_end:                       # settle
            for _ in range(8): ops.gemm(a, w, b, ops.EPI_BF16, out=out)
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        s.record()
        for _ in range(reps): ops.gemm(a, w, b, ops.EPI_BF16, out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        row.append("%s %7.1f us = %5.0f TFLOP/s" % (kind, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12))
        del a, w, out
    print("%-26s M=%d N=%d K=%d: " % (name, M, N, K) + " | ".join(row), flush=True)
