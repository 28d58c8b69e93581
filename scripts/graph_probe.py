"""Does capturing forward_prepared into a HIP graph pay at small batches?  (launch-bound check)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 54
dev = torch.device("cuda:0")
model, hp, kw = bench.build_model(dev)
batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54[:T]] * G, 200)
with torch.no_grad():
    pb = model.prepare_batch(batch)
    for _ in range(3):
        ref, _ = model.forward_prepared(pb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.forward_prepared(pb)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model.forward_prepared(pb)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pred, emb = model.forward_prepared(pb)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 10
    print(f"G={G} T={T}: eager {eager * 1e3:.2f} ms/step, graph replay {graph * 1e3:.2f} ms/step, same result: {torch.equal(pred, ref)}")
