"""Seq2GenePredictor: the older two-module layout of the same network (reference seq2gene/model.py:24-459,
seq2gene/modules/layers.py:579-1009) -- an `epigenetics_modulator` holding the CRE layers and the context embedding and
a `gene_modulator` holding the gene layers.  Selectable through `model_class` (processors/model_manager.py:80-84).
The arithmetic is the CombinedModulator's (gene layer i reads the CRE stream after CRE layer i-1, layers.py:905-921);
only the sub-module names / state-dict prefixes differ, so this class reuses the whole HIP path -- with every option of the
reference class: cross-attention-only gene layers (its constructor default), cross_alibi, use_res, context-free CRE layers
(fixtures small_twomod, small_twomod_b, small_twomod_c from the reference's own class)."""
from __future__ import annotations

import torch.nn as nn

from ..utils.constants import REF_CREs
from .model_combined_modulator import Seq2GenePredictorCombinedModulator, modulator_forward_packed
from .modules.layers import (ContextFlashAttentionEncoderLayer, ContextFlashCrossAttentionEncoderLayer,
                             FlashAttentionEncoderLayer)


class EpigeneticsModulator(nn.Module):
    """The CRE layers: with the second-level context embedding (reference layers.py:594-611) or context-free
    (use_context=False, :612-624)."""

    def __init__(self, emb_dim, num_heads, num_layers, use_alibi, mlp_dout, use_context, num_ref_cres=None, flash_attn_3=False):
        super().__init__()
        self.use_context = use_context
        if use_context:
            assert num_ref_cres is not None, "num_ref_cres must be provided when use_context is True"
            self.second_level_context_embedding = nn.Embedding(num_ref_cres, emb_dim)
            mk = lambda: ContextFlashAttentionEncoderLayer(  # noqa: E731
                d_model=emb_dim, nhead=num_heads, batch_first=True, use_alibi=use_alibi, mlp_dout=mlp_dout,
                flash_attn_3=flash_attn_3)
        else:
            self.second_level_context_embedding = None
            mk = lambda: FlashAttentionEncoderLayer(  # noqa: E731
                d_model=emb_dim, nhead=num_heads, batch_first=True, use_alibi=use_alibi, mlp_dout=mlp_dout)
        self.epigenetics_modulator = nn.ModuleList([mk() for _ in range(num_layers - 1)])


class GeneModulator(nn.Module):
    """The gene layers: cross-attention-only (the constructor default, reference layers.py:753,765-780) or self + cross
    attention (:781-796); `use_res` adds the gene-stream input back after every layer (:908-912), `cross_alibi` puts ALiBi
    on the gene -> CRE cross attention."""

    def __init__(self, emb_dim, num_heads, num_layers, use_alibi, mlp_dout, only_cross_attention=True, use_res=False,
                 cross_alibi=False, flash_attn_3=False):
        super().__init__()
        self.use_res, self.only_cross_attention, self.cross_alibi = use_res, only_cross_attention, cross_alibi
        cls = ContextFlashCrossAttentionEncoderLayer if only_cross_attention else ContextFlashAttentionEncoderLayer
        self.gene_modulator = nn.ModuleList([
            cls(d_model=emb_dim, nhead=num_heads, batch_first=True, use_alibi=use_alibi, mlp_dout=mlp_dout,
                cross_alibi=cross_alibi, flash_attn_3=flash_attn_3) for _ in range(num_layers)])


class Seq2GenePredictor(Seq2GenePredictorCombinedModulator):
    def _build_modulator(self, emb_dim, num_heads, num_layers, use_alibi, mlp_dout, use_context, flash_attn_3):
        self.epigenetics_modulator = EpigeneticsModulator(
            emb_dim=emb_dim, num_heads=num_heads, num_layers=num_layers, use_alibi=use_alibi, mlp_dout=mlp_dout,
            use_context=use_context, num_ref_cres=len(REF_CREs) if use_context else None, flash_attn_3=flash_attn_3)
        self.gene_modulator = GeneModulator(
            emb_dim=emb_dim, num_heads=num_heads, num_layers=num_layers, use_alibi=use_alibi, mlp_dout=mlp_dout,
            only_cross_attention=self.only_cross_attention, use_res=self.use_res, cross_alibi=self.cross_alibi,
            flash_attn_3=flash_attn_3)

    def _modulator_forward_packed(self, *a, **k):
        em = self.epigenetics_modulator
        return modulator_forward_packed(em.second_level_context_embedding, em.epigenetics_modulator,
                                        self.gene_modulator.gene_modulator, *a, use_res=self.gene_modulator.use_res, **k)

    def predict_step(self, batch, batch_idx, dataloader_idx=None):
        """The older class reads the CRE masks under "cre_attention_mask" (reference seq2gene/model.py:656);
        both spellings are accepted."""
        if "cre_attention_masks" not in batch and "cre_attention_mask" in batch:
            batch = dict(batch, cre_attention_masks=batch["cre_attention_mask"])
        return super().predict_step(batch, batch_idx, dataloader_idx)
