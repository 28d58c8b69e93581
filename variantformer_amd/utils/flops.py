"""Algorithmic work model of the hot path: GEMM = 2*M*K*N flop, attention = 4*sq*sk*width."""
from __future__ import annotations


def gene_flops(N: int, C: int, T: int, S_c: int, S_g: int, sum_sq_cre: float, sum_sq_gene: float,
               d: int = 512, ell: int = 6, D: int = 1536, L_layers: int = 25, d_gene: int | None = None,
               executed_by_reference: bool = False) -> dict:
    """FLOPs for one gene.  S_c / S_g = valid tokens over CRE windows / gene chunks; sum_sq_* = sum over windows
    of (valid tokens)^2.  executed_by_reference=True counts the CRE stream and the gene-layer K/V projection
    once per tissue, as the reference executes them."""
    G = C + 1
    F = 2048
    d_gene = d if d_gene is None else d_gene
    per_tok = lambda w: 8 * w * w + 3 * F * w          # Wqkv 6w^2 + out 2w^2 + geglu1 2*w*F + geglu2 2*(F/2)*w  # noqa: E731
    seq2reg = ell * (S_c * per_tok(d) + S_g * per_tok(d_gene)) + ell * 4 * (sum_sq_cre * d + sum_sq_gene * d_gene)
    if not executed_by_reference:
        # the mean pool commutes with the last layer's down-projection (F/2 -> w): algorithmically that Linear runs on one
        # pooled row per window / chunk instead of on every token (seq2reg/modules.py, _pooled_down_projection)
        seq2reg -= (S_c - N) * F * d + (S_g - C) * F * d_gene
        # the first layer's Wqkv acts on Embedding(id) + positional(position): vocab x token_length distinct rows, projected
        # once per model, looked up per token (seq2reg/model.py, _layer0_qkv_table): no per-gene arithmetic
        seq2reg -= S_c * 6 * d * d + S_g * 6 * d_gene * d_gene
    maps = 2 * N * d * D + 2 * C * d_gene * D
    Lc, Lg = L_layers - 1, L_layers
    # CRE layer per token: self 8D^2, cross Wq 2D^2 + out 2D^2 (+ Wkv on the 9-row table: negligible), FFN 3FD
    # context cross attention: the keys of a gene are copies of the 9 label embeddings, so algorithmically it is softmax over
    # the 9 distinct rows with log(count) added (vf_attn_counted_keys): 4 * N * 9 * D instead of 4 * N^2 * D
    cre_stream = Lc * (N * (12 * D * D + 3 * F * D) + 4 * N * N * D + 4 * N * 9 * D)
    cre_stream_ref = Lc * (N * (16 * D * D + 3 * F * D) + 8 * N * N * D)      # reference also projects Wkv per token
    kv_proj = Lg * N * 4 * D * D
    gene_layer = G * (12 * D * D + 3 * F * D) + 4 * G * G * D + 4 * G * N * D
    # last gene layer: only the registry row is consumed downstream (pool_outputs row 0), so algorithmically it needs
    # the K/V projection of all G tokens and everything else for ONE row
    gene_last = G * 4 * D * D + (8 * D * D + 3 * F * D) + 4 * G * D + 4 * N * D
    gene_stream = (Lg - 1) * gene_layer + (gene_layer if executed_by_reference else gene_last)
    head = 2 * (2 * D * D) + 2 * D
    if executed_by_reference:
        total = seq2reg + maps + T * (cre_stream_ref + kv_proj + gene_stream + head)
    else:
        total = seq2reg + maps + cre_stream + kv_proj + T * (gene_stream + head)
    return {"total": float(total), "seq2reg": float(seq2reg), "cre_stream": float(cre_stream), "kv_proj": float(kv_proj),
            "gene_stream": float(T * gene_stream), "head": float(T * head)}


def batch_flops(batch: dict, d=512, ell=6, D=1536, L_layers=25, executed_by_reference=False) -> float:
    tot = 0.0
    for i in range(len(batch["cre_sequences"])):
        cl = (~batch["cre_attention_masks"][i][:, 0, :]).sum(1).double()
        gl = (~batch["gene_attention_masks"][i][:, 0, :]).sum(1).double()
        tot += gene_flops(len(cl), len(gl), len(batch["tissue_context"][i]), float(cl.sum()), float(gl.sum()),
                          float((cl * cl).sum()), float((gl * gl).sum()), d, ell, D, L_layers,
                          executed_by_reference=executed_by_reference)["total"]
    return tot
