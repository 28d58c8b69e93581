"""ModelManager: same public surface as the reference's processors/model_manager.py:21-121
(`ModelManager(config).load_model() -> (model, checkpoint_path)`), building the HIP-backed model classes.

`config` is the `model:` block of configs/vf_model.yaml (an attribute-style mapping: OmegaConf if the caller
has it, otherwise variantformer_amd.utils.config.Config).  Checkpoints are the reference's Lightning-style
files: {"hyper_parameters", "state_dict"} for the tokenizers (:44-59), {"state_dict"} or a bare state dict
for the full model (:107-113)."""
from __future__ import annotations

import logging
import os
from typing import Tuple

import torch

from ..seq2gene.model import Seq2GenePredictor
from ..seq2gene.model_combined_modulator import Seq2GenePredictorCombinedModulator
from ..seq2reg.model import Seq2RegPredictor
from ..utils.config import Config

log = logging.getLogger(__name__)


class ModelManager:
    """Handles model loading and inference operations"""

    def __init__(self, config):
        self.config = config
        self.model = None
        # the reference falls back to "cpu" (:27); this build has no CPU compute path, load_model() says so
        self.device = "cuda" if torch.cuda.is_available() else "cpu"

    def _load_seq2reg_from(self, path: str) -> Seq2RegPredictor:
        chk = torch.load(path, map_location="cpu", weights_only=False)
        seq2reg = Seq2RegPredictor(**chk["hyper_parameters"])
        seq2reg.load_state_dict(chk["state_dict"])
        return seq2reg

    def _load_seq2reg(self, train_cfg) -> Seq2RegPredictor:
        return self._load_seq2reg_from(train_cfg.cre_tokenizer.path)

    def _load_seq2reg_gene(self, train_cfg) -> Seq2RegPredictor:
        return self._load_seq2reg_from(train_cfg.gene_tokenizer.path)

    def load_model(self) -> Tuple[Seq2GenePredictorCombinedModulator, str]:
        train_cfg = self.config.copy() if hasattr(self.config, "copy") else Config(dict(self.config))
        if not isinstance(train_cfg, Config):
            train_cfg = Config({k: train_cfg[k] for k in train_cfg})
        log.info("Loading Seq2Reg model...")
        seq2reg = self._load_seq2reg(train_cfg)
        log.info("Loading Seq2Reg gene model...")
        seq2reg_gene = self._load_seq2reg_gene(train_cfg)
        delattr(train_cfg, "cre_tokenizer")
        delattr(train_cfg, "gene_tokenizer")
        train_cfg.token_dim = seq2reg.hparams.embedding_dim                      # :77
        model_classes = {"Seq2GenePredictor": Seq2GenePredictor,
                         "Seq2GenePredictorCombinedModulator": Seq2GenePredictorCombinedModulator}
        name = train_cfg.get("model_class", "Seq2GenePredictor")
        if name not in model_classes:
            raise NotImplementedError(f"model_class {name!r} is not one of {sorted(model_classes)}")
        log.info("Creating Seq2Gene model...")
        gene_model = model_classes[name](cre_tokenizer=seq2reg, gene_tokenizer=seq2reg_gene, **train_cfg)
        for cname, module in gene_model.named_children():
            log.info(f"  {cname}: {sum(p.numel() for p in module.parameters()):,} params")
        log.info(f"Total number of parameters: {sum(p.numel() for p in gene_model.parameters()):,}")
        checkpoint_path = self.config.checkpoint_path
        if not os.path.exists(checkpoint_path):
            raise ValueError("Checkpoint not found")
        log.info(f"Loading checkpoint from {checkpoint_path}")
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        gene_model.load_state_dict(checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint)
        if self.device != "cuda":
            raise RuntimeError("no GPU visible: the HIP inference path has no CPU fallback")
        gene_model.eval()
        gene_model.to(self.device)
        gene_model.vep = False
        log.info(f"Model loaded successfully on {self.device}")
        self.model = gene_model
        return gene_model, checkpoint_path
