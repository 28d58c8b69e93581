"""VariantProcessor: the variant-effect (VEP) API surface (reference processors/variantprocessor.py:26-528) over the
HIP model: variants table -> (variant, gene, population) pairs -> VEPDataset ref/het/hom batches ->
`variant_prediction` -> long-format DataFrame (one row per variant x gene x tissue x population x zygosity) ->
log2 fold-change scores.

The genome artifacts (per-gene cCRE CSVs, per-population gene / cCRE sequence tables; S3 + duckdb manifests in the
reference, utils/assets.py) are passed in as lookups with the reference's `get_file_path` methods."""
from __future__ import annotations

import os
from pathlib import Path
from typing import List

import numpy as np
import pandas as pd
import torch
import yaml
from torch.utils.data import DataLoader

from ..datasets.vepdataset import VEPDataset, Variant, collate_fn
from ..utils.config import Config, load_yaml
from ..utils.functions import generate_log2fc_score
from ..utils.seq import BPEEncoder
from .model_manager import ModelManager
from .multi_datasets_loader import MultiDatasetsLoader
from .trainer import Trainer

_GENOTYPES = ("2", "1", "0")            # zygosity labels in output order: hom, het, ref
_OUTPUT_KEYS = ("pred_gene_exp", "embd", "gene_token_embedding", "cre_token_embedding")


class VariantProcessor:
    def __init__(self, model_class: str = "v4_pcg", config_dir: str | None = None, gene_cre_manifest=None,
                 gene_seq_manifest=None, cre_seq_manifest=None, require_gpu: bool = True):
        base_dir = Path(__file__).parent.parent.resolve()
        self.config_location = Path(config_dir) if config_dir else base_dir / "configs"
        model_config = load_yaml(str(self.config_location / "vf_model.yaml"))[model_class]
        vep_loader_config = load_yaml(str(self.config_location / "veploader.yaml"))
        self.gene_cre_manifest, self.gene_seq_manifest, self.cre_seq_manifest = \
            gene_cre_manifest, gene_seq_manifest, cre_seq_manifest
        root = self.config_location.parent
        for node, key in ((vep_loader_config, "CRE_BED"), (vep_loader_config, "fasta_path"), (vep_loader_config, "af_path"),
                          (model_config.dataset, "gencode_v24"), (model_config.model, "checkpoint_path"),
                          (model_config.model.cre_tokenizer, "path"), (model_config.model.gene_tokenizer, "path")):
            if not os.path.isabs(node[key]):
                node[key] = str(root / node[key])
        self.vep_loader_config = vep_loader_config
        self.config = self._create_vep_config(model_config, vep_loader_config)
        self.multi_data_loader = MultiDatasetsLoader(self.config)
        self.model_manager = ModelManager(Config(dict(model_config.model)))
        self.vep_dataset = None
        self.populations = ["REF_HG38", "EAS", "EUR", "AFR", "SAS", "AMR"]
        with open(base_dir / "vocabs" / "dataset_vocab.yaml") as f:
            self.data_vocab = Config(yaml.safe_load(f))
        with open(base_dir / "vocabs" / "tissue_vocab.yaml") as f:
            self.tissue_vocab = yaml.safe_load(f)
        self.tissue_idx_to_name = {v: k for k, v in self.tissue_vocab.items()}
        self.fasta_path = vep_loader_config.fasta_path
        self.require_gpu = require_gpu

    def _create_vep_config(self, model_config, data_config):
        d = model_config.dataset
        return Config({"all_cres": data_config.CRE_BED, "checkpoint_path": model_config.model.checkpoint_path,
                       "train_config": model_config, "gencode": d.gencode_v24, "emb_dim": model_config.model.emb_dim,
                       "cre_neighbour_hood": d.cre_neighbour_hood,
                       "gene_upstream_neighbour_hood": d.gene_upstream_neighbour_hood,
                       "gene_downstream_neighbour_hood": d.gene_downstream_neighbour_hood, "max_length": d.max_length,
                       "context_window": d.max_chunks, "fasta_path": data_config.fasta_path,
                       "precision": model_config.model.precision, "af_path": data_config.af_path})

    def _setup_output_directory(self, output_dir: str = None):
        Path(output_dir).mkdir(parents=True, exist_ok=True)

    # -- pairs ----------------------------------------------------------------------------------------
    def build_pairs(self, variants: List[Variant], vcf_path: str = None, sample_name: str = None):
        """(variant, gene, population) work list (reference :151-196): with a donor VCF the donor ("SAMPLE") and the
        hg38 reference, otherwise the six population genomes."""
        pairs, mapped = [], 0
        for variant in variants:
            genes = self.multi_data_loader.get_probable_genes(variant)
            if len(variant.gene_id) != 0:
                genes = [g for g in genes if g["gene_id"].split(".")[0] in variant.gene_id]
            mapped += len(genes) != 0
            for gene in genes:
                if vcf_path is not None and sample_name is not None:
                    pairs.append({"variant": variant, "gene": gene, "population": "SAMPLE", "sample_name": sample_name,
                                  "vcf_path": vcf_path})
                    pairs.append({"variant": variant, "gene": gene, "population": "REF_HG38", "sample_name": "hg38",
                                  "vcf_path": None})
                else:
                    for pop in self.populations:
                        pairs.append({"variant": variant, "gene": gene, "population": pop,
                                      "sample_name": self.data_vocab[pop].sample_name, "vcf_path": None})
        return pairs, mapped

    def initialize(self, var_df: pd.DataFrame, output_dir: str, vcf_path: str = None, sample_name: str = None):
        self.config.output_location = output_dir
        self._setup_output_directory(output_dir)
        self.multi_data_loader.load_annotations()
        variants = self.load_variants(var_df)
        if self._check_variant_exists():
            raise ValueError(f"Variants already processed at {self._get_variant_output_path()}. To reprocess, change "
                             f"the output directory or remove the existing file.")
        self.gene_variant_pairs, mapped = self.build_pairs(variants, vcf_path, sample_name)
        if mapped == 0:
            raise ValueError("No gene-variant pairs found. Check your input data and annotations.")
        bpe = BPEEncoder()
        bpe.load_vocabulary()
        vep_dataset = VEPDataset(
            bpe_encoder=bpe, gene_cre_manifest=self.gene_cre_manifest, gene_seq_manifest=self.gene_seq_manifest,
            cre_seq_manifest=self.cre_seq_manifest, max_length=self.config.max_length,
            context_window=self.config.context_window, cre_neighbour_hood=self.config.cre_neighbour_hood,
            gene_upstream_neighbour_hood=self.config.gene_upstream_neighbour_hood,
            gene_downstream_neighbour_hood=self.config.gene_downstream_neighbour_hood,
            gene_variant_pairs=self.gene_variant_pairs, data_vocab=self.data_vocab, fasta_path=self.fasta_path)
        cfg = dict(self.vep_loader_config.dataloader)
        kw = {"num_workers": cfg.get("num_workers", 0), "pin_memory": cfg.get("pin_memory", False)}
        if kw["num_workers"] > 0:
            kw["prefetch_factor"] = cfg.get("prefetch_factor", 2)
        dataloader = DataLoader(vep_dataset, batch_size=1, shuffle=False, collate_fn=collate_fn, **kw)
        model, ckpt_path = self.model_manager.load_model()
        trainer = Trainer(accelerator="gpu", devices=1, logger=False, precision=self.config.precision,
                          enable_checkpointing=False)
        model.vep = True
        return vep_dataset, dataloader, model, trainer, ckpt_path

    def load_variants(self, var_df: pd.DataFrame) -> List[Variant]:
        variants_df = self.multi_data_loader._load_variants(var_df)
        return self.multi_data_loader.create_variant_objects(variants_df, self.tissue_vocab)

    def _check_variant_exists(self) -> bool:
        return os.path.exists(self._get_variant_output_path())

    def _get_variant_output_path(self) -> str:
        src = self.config.variants_file if "variants_file" in self.config else "vep"
        stem = src.split("/")[-1].split(".")[0]
        if "chunks" in self.config and self.config.chunks > 1:
            return os.path.join(self.config.output_location, f"{stem}_chunk{self.config.chunk_id}_VF.parquet")
        return os.path.join(self.config.output_location, f"{stem}_VF.parquet")

    # -- outputs --------------------------------------------------------------------------------------
    def compile_predictions(self, predictions, vcf_path=None):
        """Long-format table, rows ordered pair -> tissue -> zygosity (2, 1, 0) (reference :303-445).  Pairs without
        overlap give NaN rows.  Without a donor VCF the zygosity-0 rows are kept for REF_HG38 only."""
        cols = {k: [] for k in ("chrom", "pos", "ref", "alt", "genes", "tissues", "variant_type", "population",
                                "sample_name", "zygosity", "gene_exp", "gene_emb", "gene_token_embedding",
                                "cre_token_embedding")}
        emb_dim = self.config.emb_dim
        for attr, pred in zip(self.gene_variant_pairs, predictions):
            variant, gene = attr["variant"], attr["gene"]
            n_t = len(variant.tissue)
            if len(pred["pred_gene_exp"]) == 0:
                per_key = {k: [np.full((n_t, 1 if k == "pred_gene_exp" else emb_dim), np.nan, dtype=np.float32)] * 3
                           for k in _OUTPUT_KEYS}
            else:
                per_key = {k: [pred[k][2], pred[k][1], pred[k][0]] for k in _OUTPUT_KEYS}      # hom, het, ref
            for t in range(n_t):
                for z, zyg in enumerate(_GENOTYPES):
                    cols["chrom"].append(variant.chrom)
                    cols["pos"].append(variant.pos)
                    cols["ref"].append(variant.ref)
                    cols["alt"].append(variant.alt)
                    cols["genes"].append(gene["gene_id"])
                    cols["tissues"].append(self.tissue_idx_to_name[variant.tissue[t]])
                    cols["variant_type"].append(pred["variant_type"])
                    cols["population"].append(attr["population"])
                    cols["sample_name"].append(attr["sample_name"])
                    cols["zygosity"].append(zyg)
                    cols["gene_exp"].append(per_key["pred_gene_exp"][z][t, 0])
                    cols["gene_emb"].append(per_key["embd"][z][t, :])
                    cols["gene_token_embedding"].append(per_key["gene_token_embedding"][z][t, :])
                    cols["cre_token_embedding"].append(per_key["cre_token_embedding"][z][t, :])
        df = pd.DataFrame(cols)
        if vcf_path is None:
            df = df[(df["zygosity"] != "0") | (df["population"] == "REF_HG38")].reset_index(drop=True)
        output_file = self._get_variant_output_path()
        if output_file.endswith(".csv"):
            df.to_csv(output_file, index=False)
        elif output_file.endswith(".parquet"):
            df.to_parquet(output_file)
        return df

    def eqtl_scores(self, df: pd.DataFrame):
        return generate_log2fc_score(df, self.config.af_path)

    def format_scores(self, df: pd.DataFrame):
        """Wide table: one column "<population>-<zygosity>-exp" per genotype, rows without the hg38 reference
        prediction dropped (reference :454-497)."""
        df["variant_id"] = df[["chrom", "pos", "ref", "alt"]].astype(str).agg("_".join, axis=1)
        df["gt-exp"] = df["population"] + "-" + df["zygosity"] + "-exp"
        df = df.rename(columns={"chrom": "chr"})
        index = ["variant_id", "genes", "tissues", "chr", "pos", "ref", "alt", "variant_type"]
        wide = (df[["variant_id", "genes", "tissues", "variant_type", "gt-exp", "gene_exp", "chr", "pos", "ref", "alt"]]
                .drop_duplicates(subset=["variant_id", "genes", "tissues", "variant_type", "gt-exp"], keep="first")
                .pivot(index=index, columns="gt-exp", values="gene_exp").reset_index())
        return wide.dropna(subset=["REF_HG38-0-exp"]).reset_index(drop=True)

    def predict(self, var_df: pd.DataFrame, output_dir: str, vcf_path: str = None, sample_name: str = None):
        vep_dataset, dataloader, model, trainer, ckpt_path = self.initialize(var_df, output_dir, vcf_path, sample_name)
        predictions = trainer.predict(model, dataloader)
        df = self.compile_predictions(predictions, vcf_path=vcf_path)
        self.cleanup()
        return df

    def cleanup(self):
        if getattr(self.model_manager, "model", None) is not None:
            del self.model_manager.model
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
