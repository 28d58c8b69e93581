"""Variant table and gene annotation handling of the VEP flow (reference processors/multi_datasets_loader.py:15-124)."""
from __future__ import annotations

from typing import List

import pandas as pd

from ..datasets.vepdataset import Variant

_BED_COLUMNS = ["chromosome", "start", "end", "name", "score", "strand", "thickStart", "thickEnd", "itemRgb", "type"]


class MultiDatasetsLoader:
    def __init__(self, config):
        self.config = config
        self.gencode_genes = None
        self.all_cres = None
        self.variants = None

    def load_annotations(self):
        """Gene table (CSV) and the ENCODE cCRE BED (10 columns, no header)."""
        self.gencode_genes = pd.read_csv(self.config.gencode)
        self.all_cres = pd.read_csv(self.config.all_cres, sep="\t", header=None)
        self.all_cres.columns = _BED_COLUMNS

    def _load_variants(self, var_df: pd.DataFrame) -> pd.DataFrame:
        df = var_df.rename(columns={"chr": "chrom"}) if "chr" in var_df.columns else var_df
        for column in ("chrom", "pos", "ref", "alt", "tissue"):
            if column not in df.columns:
                raise ValueError(f"Column {column} not found in {var_df.columns}")
        return df.sort_values(by=["chrom", "pos"]).reset_index(drop=True)

    def get_probable_genes(self, variant: Variant, window_size: int = 1000000) -> List[dict]:
        """Genes whose body, widened by window_size on both sides, strictly contains the variant."""
        if self.gencode_genes is None:
            raise RuntimeError("Gene annotations not loaded. Call load_annotations() first.")
        g = self.gencode_genes
        near = g[(g["chromosome"] == variant.chrom) & (g["start"] - window_size < variant.pos)
                 & (variant.pos < g["end"] + window_size)]
        return [{"gene_id": r["gene_id"], "start": r["start"], "end": r["end"], "gene_name": r["gene_name"],
                 "strand": r["strand"], "chromosome": r["chromosome"]} for _, r in near.iterrows()]

    def create_variant_objects(self, df: pd.DataFrame, tissue_vocab: dict) -> List[Variant]:
        out = []
        for _, row in df.iterrows():
            genes = row.get("gene_id", "").split(",") if "gene_id" in row else []
            out.append(Variant(chrom=row["chrom"], pos=row["pos"], ref=row["ref"], alt=row["alt"],
                               consequence=row.get("consequence", "NA"), label=row.get("label", "NA"),
                               tissue=[tissue_vocab[t] for t in row["tissue"].split(",")],
                               gene_id=[g.split(".")[0] for g in genes]))
        return out
