"""Per-launch-geometry GEMM time of one bench step (HIP events on the launch stream, ops.KernelTimer(detail=True)).
usage: python scripts/shape_breakdown.py [genes_per_step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from variantformer_amd import ops
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
model, hp, kw = bench.build_model(dev)
batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54] * G, 200)
with torch.no_grad():
    pb = model.prepare_batch(batch)
    for _ in range(2):
        model.forward_prepared(pb)
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer(detail=True)
    for _ in range(3):
        model.forward_prepared(pb)
    summ = ops.TIMER.summary()
    ops.TIMER = None
tot = summ["gemm"]["total_ms"]
print(f"GEMM total {tot / 3:.2f} ms/step, {summ['gemm']['flops'] / tot / 1e9:.0f} TFLOP/s")
rows = [(k, v) for k, v in summ.items() if k.startswith("gemm[")]
rows.sort(key=lambda kv: -kv[1]["total_ms"])
for k, v in rows:
    print(f"{k:48s} n={v['launches'] // 3:4d} {v['total_ms'] / 3:8.3f} ms/step {100 * v['total_ms'] / tot:5.1f}%  "
          f"{v['flops'] / v['total_ms'] / 1e9:7.0f} TFLOP/s  {v['bytes'] / v['total_ms'] / 1e6:7.0f} GB/s")
