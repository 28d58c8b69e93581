"""Batch format of the hot path (reference datasets/vcfdataset.py:18-63) and a synthetic per-gene dataset
with the reference's sample tuple, used by tests / bench where the genome artifacts are unavailable."""
from __future__ import annotations

import pandas as pd
import torch
from torch.utils.data import Dataset

from ..utils import synthetic


def collate_fn_batching(batch):
    """Tuples (X, mask, tissues, labels, ref_labels, strand, gene_chunks, gene_masks) -> dict of lists, as the
    reference's collate_fn_batching."""
    keys = ["cre_sequences", "cre_attention_masks", "tissue_context", "cre_labels", "ref_cre_labels",
            "gene_embeddings", "gene_attention_masks"]
    out = {k: [] for k in keys}
    strands = []
    for X, mask, ctx, label, ref_label, strand, emb, emb_att in batch:
        out["cre_sequences"].append(X)
        out["cre_attention_masks"].append(mask)
        out["tissue_context"].append(ctx)
        out["cre_labels"].append(label)
        out["ref_cre_labels"].append(ref_label)
        out["gene_embeddings"].append(emb)
        out["gene_attention_masks"].append(emb_att)
        strands.append(strand.unsqueeze(0))
    out["strand_val"] = torch.cat(strands, dim=0)
    return out


class SyntheticGeneDataset(Dataset):
    """query_df rows (gene_id, tissues as comma separated names) -> seeded synthetic samples
    (SURVEY.md §8d geometry).  Mirrors VCFDataset's query filtering (:123-169) and sample tuple (:305-336)."""

    def __init__(self, query_df: pd.DataFrame, tissue_vocab: dict, n_cre=300, n_chunks=200, token_length=200,
                 seed=20251205, **_):
        rows = []
        for _, row in query_df.iterrows():
            names = [t for t in row["tissues"].split(",") if t in tissue_vocab]
            if names:
                rows.append({"gene_id": row["gene_id"], "tissues": [tissue_vocab[t] for t in names], "tissue_names": names})
        if not rows:
            raise ValueError("No genes found in the query df with at least one known tissue")
        self.query_df = pd.DataFrame(rows)
        self.n_cre, self.n_chunks, self.L, self.seed = n_cre, n_chunks, token_length, seed

    def __len__(self):
        return len(self.query_df)

    def __getitem__(self, idx):
        r = self.query_df.iloc[idx]
        g = synthetic.make_gene(self.seed + idx, self.n_cre, self.n_chunks, r["tissues"], self.L)
        return (g["cre_sequences"], g["cre_attention_masks"], g["tissue_context"],
                torch.zeros_like(g["ref_cre_labels"]), g["ref_cre_labels"], g["strand"],
                g["gene_embeddings"], g["gene_attention_masks"])
