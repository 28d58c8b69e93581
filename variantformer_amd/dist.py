"""Multi-GPU: genes shard embarrassingly, one process per GPU, one exchange step.

The reference has no distributed code (Trainer(devices=1), processors/vcfprocessor.py:252-258).  Here every rank
holds a full weight replica, takes the genes an LPT (longest-processing-time-first) assignment gives it, runs the
single-GPU path, and ONE all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests)
reassembles the per-donor expression matrix in query order on every rank.  Messages are tiny (32 genes x 54 x 4 B
per rank at cfg 3), i.e. latency-bound; blocks are padded to the largest shard so a plain all_gather works.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def gene_cost(n_cre: int, n_chunks: int, n_tissues: int, tokens_per_cre: float = 100.0, L: int = 200) -> float:
    """Relative cost of a gene (same terms as utils.flops.gene_flops at production widths)."""
    from .utils.flops import gene_flops
    S_c, S_g = n_cre * tokens_per_cre, n_chunks * L
    return gene_flops(n_cre, n_chunks, n_tissues, S_c, S_g, n_cre * tokens_per_cre ** 2, n_chunks * L * L)["total"]


def shard_genes_lpt(costs, world_size: int) -> list[list[int]]:
    """Greedy LPT: genes sorted by decreasing cost, each given to the currently lightest rank.
    Returns, per rank, the gene indices it owns (in increasing index order).  Deterministic."""
    costs = np.asarray(costs, dtype=np.float64)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owned = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += float(costs[i])
    return [sorted(o) for o in owned]


def all_gather_expression(local: torch.Tensor, owned: list[list[int]], n_genes: int, group=None) -> torch.Tensor:
    """local: [len(owned[rank]), T(, ...)] rows for this rank's genes in owned[rank] order.
    Returns the full [n_genes, T(, ...)] tensor in query order on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert local.shape[0] == len(owned[rank])
    if not dist.is_initialized():
        out = torch.empty((n_genes,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(owned[0], device=local.device, dtype=torch.long)] = local
        return out
    max_rows = max(len(o) for o in owned)
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = torch.empty((world * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, pad.contiguous(), group=group)
    out = torch.empty((n_genes,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        if owned[r]:
            idx = torch.as_tensor(owned[r], device=local.device, dtype=torch.long)
            out[idx] = gathered[r * max_rows: r * max_rows + len(owned[r])]
    return out


def shard_batch(batch: dict, gene_ids: list[int]) -> dict:
    """Sub-batch (collate_fn_batching dict) holding only the listed genes."""
    out = {}
    for k, v in batch.items():
        if isinstance(v, list):
            out[k] = [v[i] for i in gene_ids]
        else:
            out[k] = v[torch.as_tensor(gene_ids, dtype=torch.long)]
    return out


def all_gather_ragged(rows: list, owned: list[list[int]], n_items: int, device, group=None) -> list:
    """rows[j]: numpy / tensor [T_j, ...] (fp32) for this rank's item owned[rank][j]; the leading length T_j may differ
    per item (genes queried with different tissue lists), trailing dims are common.  Returns the list of n_items arrays
    in query order on every rank.  ONE padded all-gather of the values plus one of the lengths (RCCL over xGMI with the
    "nccl" backend): blocks are padded to the largest shard and the longest item."""
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert len(rows) == len(owned[rank])
    rows = [torch.as_tensor(r, dtype=torch.float32) for r in rows]
    trail = tuple(rows[0].shape[1:]) if rows else ()
    assert len(trail) <= 4, "all_gather_ragged: at most 4 trailing dims (the metadata exchanged between ranks is fixed-size)"
    if dist.is_initialized():                       # agree on trailing dims / longest item even if a shard is empty
        meta = torch.tensor([max([r.shape[0] for r in rows], default=0)] + list(trail) + [-1] * (4 - len(trail)),
                            dtype=torch.int64, device=device)
        allmeta = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(allmeta, meta, group=group)
        t_max = int(max(int(m[0]) for m in allmeta))
        full = [m for m in allmeta if int(m[0]) > 0]
        if not trail and full:
            trail = tuple(int(v) for v in full[0][1:] if int(v) >= 0)
    else:
        t_max = max([r.shape[0] for r in rows], default=0)
    max_rows = max(len(o) for o in owned)
    vals = torch.zeros((max_rows, t_max) + trail, dtype=torch.float32, device=device)
    lens = torch.zeros((max_rows,), dtype=torch.int64, device=device)
    for j, r in enumerate(rows):
        vals[j, : r.shape[0]] = r.to(device)
        lens[j] = r.shape[0]
    if dist.is_initialized():
        g_vals = torch.empty((world * max_rows, t_max) + trail, dtype=torch.float32, device=device)
        g_lens = torch.empty((world * max_rows,), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(g_vals, vals, group=group)
        dist.all_gather_into_tensor(g_lens, lens, group=group)
    else:
        g_vals, g_lens = vals, lens
    g_vals, g_lens = g_vals.cpu().numpy(), g_lens.cpu().numpy()
    out = [None] * n_items
    for r in range(world):
        for j, item in enumerate(owned[r]):
            out[item] = np.ascontiguousarray(g_vals[r * max_rows + j, : g_lens[r * max_rows + j]])
    return out


def predict_sharded(predict_batch, dataset, collate_fn, costs=None, batch_size: int = 8, device="cpu", group=None,
                    loader_kwargs: dict | None = None):
    """vcf2exp over the ranks of the default process group: genes (dataset items) are dealt by LPT on
    `costs` (None: equal costs), each rank runs `predict_batch(batch, batch_idx) -> {"pred_gene_exp": [...],
    "embeddings": [...]}` (predict_step's contract) over ITS genes in batches of `batch_size`, and one padded gather per
    output reassembles the per-gene results in dataset order on every rank.  Without an initialised process group this
    is the single-process loop.  Returns ({"pred_gene_exp": [n items], "embeddings": [n items]}, busy_seconds)."""
    import time

    from torch.utils.data import DataLoader, Subset
    n = len(dataset)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    owned = shard_genes_lpt([1.0] * n if costs is None else costs, world)
    mine = owned[rank]
    preds, embs = [], []
    t0 = time.perf_counter()
    failure = None
    try:
        if mine:
            loader = DataLoader(Subset(dataset, mine), batch_size=batch_size, shuffle=False, collate_fn=collate_fn,
                                **(loader_kwargs or {}))
            for i, batch in enumerate(loader):
                out = predict_batch(batch, i)
                preds.extend(out["pred_gene_exp"])
                embs.extend(out["embeddings"])
        if len(preds) != len(mine):
            raise RuntimeError(f"rank {rank}: {len(preds)} predictions for {len(mine)} genes")
    except Exception as e:                  # noqa: BLE001 -- reported to every rank below, then re-raised
        failure = e
    busy = time.perf_counter() - t0
    # A rank that failed must not leave the others blocked in the gathers until the collective times out: every rank
    # learns about it through ONE tiny all-reduce (MAX of a 0/1 flag) and raises.
    if dist.is_initialized():
        flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()):
            if failure is not None:
                raise failure
            raise RuntimeError(f"predict_sharded: another rank failed (this is rank {rank}); no results were gathered")
    elif failure is not None:
        raise failure
    return {"pred_gene_exp": all_gather_ragged(preds, owned, n, device, group),
            "embeddings": all_gather_ragged(embs, owned, n, device, group)}, busy
