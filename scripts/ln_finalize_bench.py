import os, sys
sys.path.insert(0, '/root/repo') if os.path.isdir('/root/repo/variantformer_amd') else sys.path.insert(0, os.getcwd())
import torch
from variantformer_amd import ops, _lib
if os.environ.get("VF_LIB"):
    _lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["VF_LIB"]))
lib = _lib.load()
for name, M, D in [("gene8", 86832, 1536), ("s2r8", 769460, 512), ("cre8", 8192, 1536)]:
    n_parts = D // 32
    part = torch.rand((n_parts, M, 2), device="cuda")
    stats = torch.empty((M, 2), device="cuda")
    best = 1e9
    for r in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            lib.vf_ln_finalize2(part.data_ptr(), M, n_parts, D, 1e-5, 1.0, 8.0, 0.0, 0, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e.record(); torch.cuda.synchronize()
        if r: best = min(best, s.elapsed_time(e) / 10)
    print("%-6s M=%7d D=%5d  %7.1f us  %6.2f TB/s  checksum %.6f" % (name, M, D, best * 1e3, part.numel() * 4 / best / 1e9, float(stats.double().sum())))
