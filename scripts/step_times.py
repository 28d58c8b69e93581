"""Per-step wall time and allocator counters of the bench step, step by step (is the first timed step representative?).
usage: python scripts/step_times.py [genes_per_step] [steps] [keep]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
KEEP = len(sys.argv) > 3 and sys.argv[3] == "keep"
out = None
if os.environ.get("VF_EXPANDABLE") == "1":          # experiment: torch's expandable segments (one growing mapping per stream)
    torch._C._accelerator_setAllocatorSettings("expandable_segments:True")
dev = torch.device("cuda:0")
model, hp, kw = bench.build_model(dev)
batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54] * G, 200)
with torch.no_grad():
    pb = model.prepare_batch(batch)
    for i in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if KEEP:                      # bench.py's step: the previous step's result stays referenced while the next one runs
            pred, emb = model.forward_prepared(pb)
            out = (pred.cpu(), emb)
        else:
            model.forward_prepared(pb)[0].cpu()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = torch.cuda.memory_stats(dev)
        print(f"G={G}{' keep' if KEEP else ''} step {i:2d}: {dt * 1e3:8.2f} ms   device_allocs {st['num_device_alloc']:4d} frees {st['num_device_free']:4d} "
              f"reserved {st['reserved_bytes.all.current'] / 2**30:6.2f} GiB  peak allocated {st['allocated_bytes.all.peak'] / 2**30:6.2f} GiB"
              f"  retries {st['num_alloc_retries']}", flush=True)
