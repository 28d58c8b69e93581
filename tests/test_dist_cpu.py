"""Gene sharding + all-gather of the expression matrix on 2 processes (gloo, CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from variantformer_amd.dist import all_gather_expression, gene_cost, shard_batch, shard_genes_lpt
from variantformer_amd.utils.synthetic import cfg3_gene_sizes, make_batch


def test_lpt_is_a_balanced_partition():
    n, c = cfg3_gene_sizes(256)
    costs = [gene_cost(int(a), int(b), 54) for a, b in zip(n, c)]
    owned = shard_genes_lpt(costs, 8)
    flat = sorted(i for o in owned for i in o)
    assert flat == list(range(256))                                    # every gene exactly once
    loads = [sum(costs[i] for i in o) for o in owned]
    assert max(loads) / (sum(loads) / 8) < 1.05                        # within 5 % of perfect balance
    assert shard_genes_lpt(costs, 8) == owned                          # deterministic
    assert shard_genes_lpt([3.0], 4) == [[0], [], [], []]


def test_shard_batch_keeps_format():
    b = make_batch(1, [3, 4, 5], [2, 2, 3], [[7], [8, 9], [10]], 16, cre_len_range=(2, 10))
    s = shard_batch(b, [2, 0])
    assert len(s["cre_sequences"]) == 2 and s["cre_sequences"][0].shape[0] == 5 and s["strand_val"].shape == (2, 1)
    assert s["tissue_context"][1].tolist() == [7]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_genes, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = [float((i * 7) % 11 + 1) for i in range(n_genes)]
        owned = shard_genes_lpt(costs, world)
        # a rank's "prediction" for gene g, tissue t is g*100+t: the gathered matrix must be in query order
        local = torch.tensor([[g * 100.0 + t for t in range(T)] for g in owned[rank]], dtype=torch.float32).reshape(len(owned[rank]), T)
        full = all_gather_expression(local, owned, n_genes)
        emb = all_gather_expression(local.unsqueeze(-1).repeat(1, 1, 3), owned, n_genes)
        q.put((rank, full.numpy(), emb.shape))
    finally:
        dist.destroy_process_group()


def test_all_gather_expression_world2():
    world, n_genes, T = 2, 7, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_genes, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([[g * 100.0 + t for t in range(T)] for g in range(n_genes)], dtype=np.float32)
    for rank, full, emb_shape in res:
        np.testing.assert_array_equal(full, want)
        assert tuple(emb_shape) == (n_genes, T, 3)


def test_all_gather_single_process_path():
    owned = [[2, 0, 1]]
    local = torch.tensor([[2.0], [0.0], [1.0]])
    out = all_gather_expression(local, owned, 3)
    assert out.ravel().tolist() == [0.0, 1.0, 2.0]
