"""Synthetic genome artifacts for the VEP sample-builder parity test.

One spec (tests/golden/vep.json) -> the three on-disk artifact kinds the reference's VEPDataset reads
(datasets/vepdataset.py:171-276): per-gene CRE CSV, per-chromosome gzip-pickled CRE sequence table, per-gene
sequence .npz.  Used twice with the same spec: by tests/golden/make_golden.py to feed the reference, and by
tests/test_vepdataset_cpu.py to feed variantformer_amd.datasets.vepdataset.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd

COMP = str.maketrans("ACGTacgtNnRYSWKMryswkm", "TGCAtgcaNnYRSWMKyrswmk")

SETTINGS = dict(max_length=20, context_window=6, cre_neighbour_hood=10, gene_upstream_neighbour_hood=100,
                gene_downstream_neighbour_hood=2500)


def make_spec(seed: int = 4242):
    from variantformer_amd.utils.synthetic import randint

    n = 6000
    r = randint(n, 0, 1000, seed, 0)
    genome = np.array(list("ACGT"))[r % 4]
    genome[(r >= 990)] = "N"                       # isolated Ns
    genome[1500:1520] = "N"                        # an N run inside the + gene
    for pos in (2060, 730, 1075, 1300, 2050, 2595, 5900, 1100, 4320, 5430, 5150, 3600, 3350, 100):
        if genome[pos - 1] == "N":                 # variant sites carry a base (the N-run case below is deliberate)
            genome[pos - 1] = "ACGT"[pos % 4]
    genome = "".join(genome)
    genome = genome[:2000] + genome[2000:2100].lower() + genome[2100:]      # soft-masked stretch
    genes = [
        {"gene_id": "ENSG_PLUS", "chromosome": "chr1", "start": 1000, "end": 2600, "strand": "+"},
        {"gene_id": "ENSG_MINUS", "chromosome": "chr1", "start": 3200, "end": 5200, "strand": "-"},
    ]
    names = ["PLS", "pELS", "dELS", "CTCF-only,CTCF-bound", "DNase-H3K4me3", "pELS,CTCF-bound", "not-a-class"]
    cres = {
        # first CRE lies upstream of the gene window
        "ENSG_PLUS": [(700, 760), (1040, 1110), (1490, 1560), (2030, 2080), (2500, 2590)],
        "ENSG_MINUS": [(3300, 3390), (4300, 4345), (5100, 5180), (5400, 5460)],
    }
    cre_rows = {g: [(s, e, names[(k + 3 * (g == "ENSG_MINUS")) % len(names)]) for k, (s, e) in enumerate(v)]
                for g, v in cres.items()}
    variants = [
        # (gene, pos(1-based), alt, tissues, sample_name, population)
        ("ENSG_PLUS", 730, "T", [7, 20], "hg38", "REF_HG38"),        # CRE only (upstream of the gene window)
        ("ENSG_PLUS", 1075, "G", [3], "hg38", "REF_HG38"),           # CRE + gene
        ("ENSG_PLUS", 1300, "C", [1, 2, 3], "hg38", "REF_HG38"),     # gene only
        ("ENSG_PLUS", 2050, "A", [5], "hg38", "REF_HG38"),           # soft-masked ref allele -> het code N -> error
        ("ENSG_PLUS", 2595, "A", [5], "hg38", "REF_HG38"),           # past the last chunk -> clamped chunk index
        ("ENSG_PLUS", 5900, "A", [5], "hg38", "REF_HG38"),           # no overlap
        ("ENSG_PLUS", 1100, "A", [9], "sampleX", "EUR"),             # non-hg38 sample: no reference check; EUR
        ("ENSG_MINUS", 4320, "C", [9, 10], "sampleX", "EUR"),        # table is unsorted -> merge branch of _map_files
        ("ENSG_MINUS", 5430, "C", [11, 12], "hg38", "REF_HG38"),     # CRE only, minus strand
        ("ENSG_MINUS", 5150, "T", [0], "hg38", "REF_HG38"),          # CRE + gene, minus strand
        ("ENSG_MINUS", 3600, "G", [53], "hg38", "REF_HG38"),         # gene only, minus strand
        ("ENSG_MINUS", 3350, "A", [4], "hg38", "REF_HG38"),          # CRE + gene near the 3' end
        ("ENSG_MINUS", 100, "A", [4], "hg38", "REF_HG38"),           # no overlap
        ("ENSG_PLUS", 2060, "G", [5, 6], "sampleX", "EUR"),          # soft-masked stretch with an upper-case ref allele
        ("ENSG_PLUS", 1510, "A", [4], "sampleX", "EUR"),             # inside the N run: the position has no token
    ]
    out = []
    for g, pos, alt, tissues, sample, pop in variants:
        ref = genome[pos - 1].upper() if sample == "hg38" else "A"
        if sample == "hg38" and genome[pos - 1] != ref:              # soft-masked: the reference asserts equality
            ref = genome[pos - 1]
        if alt == ref:
            alt = "ACGT"[("ACGT".index(alt) + 1) % 4]
        out.append({"gene": g, "chrom": "1", "pos": pos, "ref": ref, "alt": alt, "tissue": tissues,
                    "sample_name": sample, "population": pop})
    return {"seed": seed, "genome": genome, "genes": genes, "cres": {g: [list(x) for x in v] for g, v in cre_rows.items()},
            "variants": out, "settings": SETTINGS, "populations": ["REF_HG38", "EUR"]}


def gene_window(gene, s):
    start, end = gene["start"], gene["end"]
    if gene["strand"] == "-":
        return max(start, end - s["gene_downstream_neighbour_hood"]), end + s["gene_upstream_neighbour_hood"]
    return max(0, start - s["gene_upstream_neighbour_hood"]), min(end, start + s["gene_downstream_neighbour_hood"])


def write_artifacts(spec, root, encode_pair):
    """encode_pair(fwd, rev) -> (ids_fwd, ids_rev).  Returns ({gene_id: csv}, {(gene_id, pop): npz},
    {(chrom, pop): pkl}) path tables for the manifest lookups."""
    s = spec["settings"]
    genome = spec["genome"]
    nb = s["cre_neighbour_hood"]
    os.makedirs(root, exist_ok=True)
    gene_csv, gene_npz, cre_pkl = {}, {}, {}
    all_rows = []
    for gene in spec["genes"]:
        rows = [{"chromosome": gene["chromosome"], "start": gene["start"], "end": gene["end"],
                 "gene_id": gene["gene_id"], "strand": gene["strand"], "start_cre": a, "end_cre": b, "cre_name": name}
                for a, b, name in spec["cres"][gene["gene_id"]]]
        path = os.path.join(root, f"{gene['gene_id']}_cres.csv")
        pd.DataFrame(rows).to_csv(path, index=False)
        gene_csv[gene["gene_id"]] = path
        all_rows += rows
        lo, hi = gene_window(gene, s)
        fwd = genome[lo:hi]
        for pop in spec["populations"]:
            p = os.path.join(root, f"{gene['gene_id']}_{pop}.npz")
            np.savez(p, sequence=fwd + "," + fwd[::-1].translate(COMP), strand=gene["strand"])
            gene_npz[(gene["gene_id"], pop)] = p
    all_rows.sort(key=lambda r: r["start_cre"])
    for pop in spec["populations"]:
        table = []
        # REF_HG38: sorted by start (contiguous-slice branch of _map_files); others: rotated (merge branch)
        order = all_rows if pop == "REF_HG38" else all_rows[4:] + all_rows[:4]
        for r in order:
            a, b = r["start_cre"] - nb, r["end_cre"] + nb
            fwd = genome[a:b]
            rev = fwd[::-1].translate(COMP)
            ids_f, ids_r = encode_pair(fwd, rev)
            table.append({"chrom": r["chromosome"], "start": a, "end": b, "cCRE": r["cre_name"],
                          f"{pop}_sequence": fwd + "," + rev,
                          f"{pop}_encoded_seq": [[float(x) for x in ids_f], [float(x) for x in ids_r]]})
        p = os.path.join(root, f"chr1_{pop}_cres.pkl.gz")
        pd.DataFrame(table).to_pickle(p, compression="gzip")
        cre_pkl[("chr1", pop)] = p
    return gene_csv, gene_npz, cre_pkl


def flatten_batch(batch):
    """dict of lists of tensors / tensors / str -> {key: ndarray} (+ variant_type)."""
    import torch

    out = {}
    for k, v in batch.items():
        if isinstance(v, str):
            continue
        if isinstance(v, torch.Tensor):
            out[k] = v.numpy()
        else:
            for j, t in enumerate(v):
                out[f"{k}.{j}"] = t.numpy()
    return out


# ---- VariantProcessor output stage (compile_predictions / format_scores / log2fc scores) ----------------------------
def make_vp_case(variant_cls, seed=77, emb_dim=6, with_sample=False):
    """Seeded (gene_variant_pairs, predictions, allele-frequency tables) for the output stage of the VEP flow."""
    rng = np.random.default_rng(seed)
    variants = [variant_cls(chrom="1", pos=1075, ref="A", alt="G", tissue=[7, 20], gene_id=["ENSG_PLUS"]),
                variant_cls(chrom="chr2", pos=3350, ref="C", alt="T", tissue=[3], gene_id=[]),
                variant_cls(chrom="1", pos=99, ref="G", alt="C", tissue=[5, 6, 9], gene_id=[])]
    genes = [{"gene_id": "ENSG_PLUS"}, {"gene_id": "ENSG_MINUS"}]
    pops = [("SAMPLE", "donor1"), ("REF_HG38", "hg38")] if with_sample else \
        [("REF_HG38", "hg38"), ("EAS", "HG00404"), ("EUR", "HG00096"), ("AFR", "HG01879"), ("SAS", "HG01589"), ("AMR", "HG00551")]
    pairs, preds = [], []
    for vi, v in enumerate(variants):
        for gi, g in enumerate(genes[: 2 if vi == 0 else 1]):
            for pop, sample in pops:
                pairs.append({"variant": v, "gene": g, "population": pop, "sample_name": sample,
                              "vcf_path": "x.vcf.gz" if pop == "SAMPLE" else None})
                n_t = len(v.tissue)
                if vi == 2 and pop in ("EAS", "SAMPLE"):          # a pair without overlap: empty prediction
                    preds.append({"pred_gene_exp": [], "embd": [], "variant_type": "No overlap",
                                  "gene_token_embedding": [], "cre_token_embedding": []})
                    continue
                p = {"variant_type": ["CRE overlap only", "Gene overlap only", "Gene and CRE overlap"][(vi + gi) % 3]}
                p["pred_gene_exp"] = [rng.random((n_t, 1), dtype=np.float32) * 5 for _ in range(3)]
                for k in ("embd", "gene_token_embedding", "cre_token_embedding"):
                    p[k] = [rng.standard_normal((n_t, emb_dim)).astype(np.float32) for _ in range(3)]
                preds.append(p)
    af = {"chr1": pd.DataFrame({"chr": ["chr1", "chr1"], "pos": [1075, 99], "ref": ["A", "G"], "alt": ["G", "C"],
                                "AF_EUR": ["0.10", "."], "AF_AFR": ["0.30", "0.0"], "AF_EAS": ["0.05", "0.0"],
                                "AF_SAS": [".", "0.0"], "AF_AMR": ["0.20", "0.0"]}),
          "chr2": pd.DataFrame({"chr": ["chr2"], "pos": [3350], "ref": ["C"], "alt": ["T"], "AF_EUR": ["0.5"],
                                "AF_AFR": ["0.25"], "AF_EAS": ["0.125"], "AF_SAS": ["0.0625"], "AF_AMR": ["0.0625"]})}
    return pairs, preds, af


def write_af_tables(af, root):
    os.makedirs(root, exist_ok=True)
    for chrom, df in af.items():
        df.to_csv(os.path.join(root, f"1KG_hg38_af_{chrom}.tsv"), sep="\t", index=False)
    return root


def frame_to_arrays(df, prefix):
    """DataFrame -> {prefix.col: ndarray} (object columns of arrays are stacked; strings kept as str arrays)."""
    out = {f"{prefix}.__columns__": np.array([str(c) for c in df.columns])}
    for c in df.columns:
        col = df[c]
        if len(col) and isinstance(col.iloc[0], np.ndarray):
            out[f"{prefix}.{c}"] = np.stack(list(col))
        elif col.dtype == object:
            out[f"{prefix}.{c}"] = np.array([str(v) for v in col])
        else:
            out[f"{prefix}.{c}"] = col.to_numpy()
    return out
