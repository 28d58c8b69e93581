"""A/B of two builds of the library on the bench step: python scripts/ab_lib.py libvf_hip_prev.so [genes_per_step] [reps]
(loads the named library from variantformer_amd/csrc/ instead of libvf_hip.so, then times the step like scripts/step_times.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variantformer_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "libvf_hip.so"
_lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), name))
import bench
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
model, hp, kw = bench.build_model(dev)
batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54] * G, 200)
ts = []
with torch.no_grad():
    pb = model.prepare_batch(batch)
    for i in range(N + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.forward_prepared(pb)[0].cpu()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
ts = ts[3:]
print(f"{name}: {G} genes per step, {min(ts):.2f} ms best, {sum(ts) / len(ts):.2f} ms mean of {len(ts)} -> {G / (sum(ts) / len(ts)) * 1e3:.2f} genes/s; checksum {float(out.double().sum()):.6f}")
