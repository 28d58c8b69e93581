"""VCFProcessor: the vcf2exp API surface (reference processors/vcfprocessor.py:23-277) over the HIP model.

Kept: constructor(model_class), get_tissues, get_genes, create_data, load_model, predict, format_output with
the same return types.  The genome-side inputs (FASTA, VCF, per-gene cCRE manifests from S3, samtools /
bcftools) are outside the hot path (SURVEY.md §2 "next"); `create_data` therefore takes a dataset factory:
anything that yields the reference's per-gene sample tuple works, e.g. datasets.SyntheticGeneDataset here or
the reference's own VCFDataset when its artifacts are available."""
from __future__ import annotations

import os
from pathlib import Path

import pandas as pd
import torch
import yaml
from torch.utils.data import DataLoader

from ..datasets.vcfdataset import collate_fn_batching
from ..utils.config import Config, load_yaml
from .model_manager import ModelManager
from .trainer import Trainer


class VCFProcessor:
    def __init__(self, model_class: str = "v4_pcg", config_dir: str | None = None, require_gpu: bool = True):
        base_dir = Path(__file__).parent.parent.resolve()
        self.config_location = Path(config_dir) if config_dir else base_dir / "configs"
        self.model_config = load_yaml(str(self.config_location / "vf_model.yaml"))[model_class]
        with open(base_dir / "vocabs" / "tissue_vocab.yaml") as f:
            self.tissue_vocab = yaml.safe_load(f)
        self.vcf_loader_config = load_yaml(str(self.config_location / "vcfloader.yaml"))
        root = self.config_location.parent
        for node, key in ((self.model_config.dataset, "gencode_v24"), (self.model_config.model, "checkpoint_path"),
                          (self.model_config.model.cre_tokenizer, "path"), (self.model_config.model.gene_tokenizer, "path")):
            if not os.path.isabs(node[key]):
                node[key] = str(root / node[key])
        if require_gpu:
            assert torch.cuda.is_available(), "GPU is not available"          # reference :60
        self.accelerator = "gpu"

    def get_tissues(self):
        return self.tissue_vocab.keys()

    def get_genes(self):
        return pd.read_csv(self.model_config.dataset.gencode_v24)

    def create_data(self, vcf_path, query_df: pd.DataFrame, dataset_factory=None, **kwargs):
        cfg = Config(dict(self.vcf_loader_config.dataloader))
        cfg.update(kwargs)
        if dataset_factory is None:
            raise NotImplementedError(
                "building samples from FASTA/VCF needs the reference's genome artifacts and samtools/bcftools "
                "(out of the hot path); pass dataset_factory=... returning a dataset with .query_df")
        dataset = dataset_factory(vcf_path=vcf_path, query_df=query_df, tissue_vocab=self.tissue_vocab,
                                  dataset_config=self.model_config.dataset)
        if cfg.get("num_workers", 0) == 0:
            cfg.pop("prefetch_factor", None)
        return dataset, DataLoader(dataset, collate_fn=collate_fn_batching, **cfg)

    def load_model(self):
        model, checkpoint_path = ModelManager(self.model_config.model).load_model()
        trainer = Trainer(accelerator=self.accelerator, devices=1, logger=False,
                          precision=self.model_config.model.precision, enable_checkpointing=False)
        return model, checkpoint_path, trainer

    def predict(self, model, checkpoint_path, trainer, dataloader, vcf_dataset):
        predictions = trainer.predict(model, dataloader, ckpt_path=checkpoint_path)
        return self.format_output(vcf_dataset.query_df, predictions)

    def format_output(self, df, predictions):
        pred_exp, embd = [], []
        for p in predictions:
            pred_exp.extend(p["pred_gene_exp"])
            embd.extend(p["embeddings"])
        pred_df = pd.DataFrame({"predicted_expression": pred_exp, "embeddings": embd})
        assert len(df) == len(pred_df), "DataFrame and predictions length mismatch"
        df["predicted_expression"] = pred_df["predicted_expression"]
        df["embeddings"] = pred_df["embeddings"]
        return df
