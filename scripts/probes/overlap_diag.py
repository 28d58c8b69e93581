"""Diagnostic (round 6): does a gene's result depend on its batch neighbours with the CRE side stream on / off?  Prints the largest
relative difference between two partitions of the first genes of BASELINE configs[2] and the self-healing bookkeeping."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from variantformer_amd import runtime
from variantformer_amd.seq2gene.model_combined_modulator import heal_state
from variantformer_amd.utils.synthetic import TISSUES_54, cfg3_gene_sizes, collate, make_gene

n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, hp, kw = bench.build_model(torch.device("cuda", 0))
n, c = cfg3_gene_sizes(256)

def gene_batch(ids):
    return collate([make_gene(20251205 * 1000003 + g, int(n[g]), int(c[g]), TISSUES_54, 200) for g in ids])

def run(partition, bs):
    expr = np.full((n_genes, 54), np.nan, np.float32)
    for shard in partition:
        for s in range(0, len(shard), bs):
            ids = shard[s:s + bs]
            out = model.predict_step(gene_batch(ids), 0)
            for j, g in enumerate(ids):
                expr[g] = out["pred_gene_exp"][j][:, 0]
    return expr

p1 = [list(range(r, n_genes, 4)) for r in range(4)]
p2 = [list(range(r * (n_genes // 4), (r + 1) * (n_genes // 4))) for r in range(4)]
import variantformer_amd.seq2gene.model_combined_modulator as M
for overlap, serialize in ((False, False), (True, False), (True, True), (True, False), (True, True)):
    M._OVERLAP_SERIALIZE_FOR_DIAG = serialize
    with runtime.override(overlap_cre_stream=overlap):
        a = run(p1, 8)
        b = run(p2, 5)
        a2 = run(p1, 8)
    hs = heal_state(model)
    print(f"overlap={overlap} serialized={serialize}: partitions differ by {float((np.abs(a - b) / np.abs(b)).max()):.2e} (relative), the same partition twice by "
          f"{float((np.abs(a - a2) / np.abs(a)).max()):.2e}; heal: finished {hs.finished} recomputed {hs.batches} off {hs.off}", flush=True)
