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
