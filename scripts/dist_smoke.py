"""RCCL smoke on however many ranks torch.distributed.run starts (the dev pool has 1 GPU per box: world size 1 still goes
through ncclCommInit, all_gather_into_tensor and the LPT bookkeeping):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scripts/dist_smoke.py
Checks variantformer_amd.dist.predict_sharded / all_gather_ragged / all_gather_expression on the nccl backend."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from tests.test_dist_cpu import _ToyGenes, _toy_collate, _toy_predict
from variantformer_amd.dist import all_gather_expression, predict_sharded, shard_genes_lpt

n = 13
res, busy = predict_sharded(_toy_predict, _ToyGenes(n), _toy_collate, costs=[float(i % 5 + 1) for i in range(n)], batch_size=4,
                            device=torch.device("cuda", local))
ds = _ToyGenes(n)
for i in range(n):
    assert res["pred_gene_exp"][i].tolist() == [[i * 1000.0 + t] for t in ds[i]["tissues"]]
    assert res["embeddings"][i].shape == (1 + i % 4, 3)
owned = shard_genes_lpt([1.0] * n, dist.get_world_size())
local_rows = torch.tensor([[g * 10.0 + t for t in range(5)] for g in owned[dist.get_rank()]], device="cuda").reshape(-1, 5)
full = all_gather_expression(local_rows, owned, n)
assert torch.equal(full.cpu(), torch.tensor([[g * 10.0 + t for t in range(5)] for g in range(n)]))
if dist.get_rank() == 0:
    print(f"dist smoke ok on {dist.get_world_size()} rank(s), backend {dist.get_backend()}, busy {busy:.4f} s")
dist.destroy_process_group()
