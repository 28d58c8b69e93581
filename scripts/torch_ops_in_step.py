"""Which torch (ATen) kernels run inside one forward of the headline step, with input shapes: everything that is not a
libvf_hip.so kernel (index plumbing, casts, copies).  usage: python scripts/torch_ops_in_step.py [genes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import ProfilerActivity, profile
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model, hp, kw = bench.build_model(dev)
batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54] * G, 200)
with torch.no_grad():
    pb = model.prepare_batch(batch)
    for _ in range(2):
        model.forward_prepared(pb)[0].cpu()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        model.forward_prepared(pb)[0].cpu()
        torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f"ATen ops with device time in one forward of {G} genes: {tot / 1e3:.3f} ms in total")
for e in rows[:25]:
    print(f"{e.key:32s} calls={e.count:4d} device_ms={e.device_time_total / 1e3:8.3f} shapes={str(e.input_shapes)[:110]}")
