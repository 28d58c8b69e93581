"""Small helpers with the reference's semantics (utils/functions.py)."""
import torch


def precision2dtype(precision_str: str) -> torch.dtype:
    """Lightning precision string -> torch dtype (reference utils/functions.py:12-32): any string containing
    'bf16' -> bfloat16, else '16' -> float16, else '32' -> float32, otherwise ValueError."""
    s = precision_str.lower().strip()
    if "bf16" in s:
        return torch.bfloat16
    if "16" in s:
        return torch.float16
    if "32" in s:
        return torch.float32
    raise ValueError(f"Unknown precision string: {precision_str}")


_COMPLEMENT = str.maketrans("ACGTRYSWKMBDHVNacgtryswkmbdhvn", "TGCAYRSWMKVHDBNtgcayrswmkvhdbn")


def reverse_complement(seq: str) -> str:
    """IUPAC-aware reverse complement (reference utils/functions.py:129-172)."""
    return seq.translate(_COMPLEMENT)[::-1]


# ---- variant-effect scores (reference utils/functions.py:184-301) ---------------------------------------------------
_POPULATIONS = ("AFR", "AMR", "EAS", "EUR", "SAS")


def merge_pop_stat(df, af_path):
    """Left-join the per-chromosome 1000-Genomes allele-frequency tables `1KG_hg38_af_<chr>.tsv` on
    (chr, pos, ref, alt); "." frequencies become NaN."""
    import os
    import numpy as np
    import pandas as pd
    parts = []
    for chrom in df["chr"].unique():
        af = pd.read_csv(os.path.join(af_path, f"1KG_hg38_af_{chrom}.tsv"), sep="\t")
        parts.append(df[df["chr"] == chrom].merge(af, on=["chr", "pos", "ref", "alt"], how="left").reset_index(drop=True))
    out = pd.concat(parts, ignore_index=True)
    for pop in ("EUR", "AFR", "EAS", "SAS", "AMR"):
        out["AF_" + pop] = out["AF_" + pop].replace(".", np.nan).astype(float)
    return out


def gene_pop_agg_score(df, score_cols, score_type="log2fc"):
    """Allele-frequency weighted mean of the population scores per row (NaN scores dropped; plain mean when every
    weight is zero; NaN when no score is valid).  The REF_HG38 homozygous column does not take part."""
    import numpy as np
    if f"VF-REF_HG38-2-exp-{score_type}" in score_cols:
        score_cols = [c for c in score_cols if "REF_HG38-2" not in c]
    af_cols = ["AF_" + c.split("-")[1] for c in score_cols if any(c.startswith(f"VF-{p}-2") for p in _POPULATIONS)]
    agg = []
    for _, row in df.iterrows():
        scores = row[score_cols].values.astype(float)
        af = row[af_cols].values.astype(float)
        ok = ~np.isnan(scores)
        if ok.sum() == 0:
            agg.append(np.nan)
            continue
        w = af[ok] / sum(af[ok])
        agg.append(np.average(scores[ok], weights=w) if np.sum(w) > 0 else np.mean(scores[ok]))
    df["VF-agg-" + score_type + "-weighted"] = agg
    return df


def generate_log2fc_score(df, af_path):
    """log2((hom + 1e-10) / (hg38 ref + 1e-10)) per population column of VariantProcessor.format_scores' wide table;
    population runs also get the allele-frequency weighted aggregate."""
    import numpy as np
    heads = tuple(f"{p}-2" for p in _POPULATIONS + ("REF_HG38", "SAMPLE"))
    pop_columns = [c for c in df.columns if c.startswith(heads)]
    keys = ["variant_id", "genes", "tissues", "ref", "alt", "chr", "pos"]
    df = df[["REF_HG38-0-exp"] + pop_columns + keys].reset_index(drop=True)
    ref = np.array(df[["REF_HG38-0-exp"]]).flatten()
    score_cols = []
    for col in pop_columns:
        df["VF-" + col + "-log2fc"] = np.log2((np.array(df[col]).flatten() + 1e-10) / (ref + 1e-10))
        score_cols.append("VF-" + col + "-log2fc")
    df[score_cols] = df[score_cols].astype(float)
    if not any(c.startswith("SAMPLE-2") for c in pop_columns):
        df = gene_pop_agg_score(merge_pop_stat(df, af_path), score_cols, score_type="log2fc")
        return df[keys + ["VF-agg-log2fc-weighted"] + score_cols]
    return df[keys + score_cols]
