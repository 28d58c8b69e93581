"""Seq2RegPredictor: the DNA -> regulatory-element encoder ("tokenizer") on the HIP kernels.

Same constructor arguments, hyper-parameter record and state-dict keys as the reference
(seq2reg/model.py:40-191) so that `Seq2RegPredictor(**chk["hyper_parameters"])` followed by
`load_state_dict(chk["state_dict"])` works unchanged (processors/model_manager.py:44-51).
Only the inference path the hot loop uses is implemented: forward(only_embed=True)
(seq2reg/model.py:193-279).  Training / classification heads are out of scope.
"""
from __future__ import annotations

import math
import types

import torch
import torch.nn as nn

from .. import ops, runtime
from .modules import FlashTransformerLayer


# Two exact re-orderings live here, both switchable for the tests through variantformer_amd.runtime.override:
#   pool_before_down_projection -- the encoder's mean pool is taken BEFORE the last layer's down-projection
#                                  (FlashTransformerLayer._pooled_down_projection); off: pool the token rows of the last layer;
#   layer0_qkv_table            -- the first layer's LayerNorm1 -> Wqkv is looked up per distinct input row
#                                  (Seq2RegPredictor._layer0_qkv_table); off: project every token.
LAYER0_QKV_TABLE_MAX_BYTES = 1 << 30         # cap on vocab * token_length rows of 3 d 16-bit values, per tokenizer and operand
                                             # type (shipped: 500 x 200 rows x 3 KB = 307 MB); above it every token is projected


def positionalencoding1d(d_model: int, length: int) -> torch.Tensor:
    """Sinusoidal table [length, d_model]; same values as the reference (seq2reg/model.py:15-37)."""
    if d_model % 2 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dim (got dim={:d})".format(d_model))
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


class Seq2RegPredictor(nn.Module):
    def __init__(self, vocab_size: int, embedding_dim: int, num_heads: int, num_layers: int, num_tissues: int,
                 num_classes: int, learning_rate: float = 1e-4, loss_fn=("cross_entropy", "0"), seq_pool: str = "mean",
                 cre_type: str = "multi", token_length: int = None, use_context: bool = False,
                 positional_encoding: str = "sinusoidal", use_flash: bool = False, majority_weight: float = None,
                 weight_decay: float = 0.0, lr_scale: float = 1.0, strand_agg: str = "mean",
                 expand_context: bool = False, mlp_dout: float = 0.1, tissues: list = None, **kwargs):
        super().__init__()
        hp = dict(vocab_size=vocab_size, embedding_dim=embedding_dim, num_heads=num_heads, num_layers=num_layers,
                  num_tissues=num_tissues, num_classes=num_classes, learning_rate=learning_rate, loss_fn=loss_fn,
                  seq_pool=seq_pool, cre_type=cre_type, token_length=token_length, use_context=use_context,
                  positional_encoding=positional_encoding, use_flash=use_flash, majority_weight=majority_weight,
                  weight_decay=weight_decay, lr_scale=lr_scale, strand_agg=strand_agg, expand_context=expand_context,
                  mlp_dout=mlp_dout, tissues=tissues)
        hp.update(kwargs)
        self.hparams = types.SimpleNamespace(**hp)
        assert use_flash, "Only Flash is supported"
        assert positional_encoding in ["sinusoidal", "alibi"], "Position encoding must be either 'sinusoidal' or 'alibi'"
        assert seq_pool in ["mean", "max", "linear"]
        self.token_embedding = nn.Embedding(vocab_size, embedding_dim)
        self.pos_encoding_type = positional_encoding
        self.seq_pool = seq_pool
        self.use_context = use_context
        self.expand_context_type = expand_context
        self.strand_agg = strand_agg
        self.num_classes = num_classes
        self.tissues = tissues
        use_alibi = positional_encoding == "alibi"
        if use_context:                       # reference seq2reg/model.py:90-95
            from ..utils.constants import REF_CREs
            self.context_embedding = nn.Embedding(len(REF_CREs), embedding_dim)
            if expand_context:
                self.expand_context = nn.Linear(1, token_length)
        if not use_alibi:
            # plain attribute like the reference (not in the state dict, :105); a device copy is cached
            self.position_encoding = positionalencoding1d(embedding_dim, token_length)
        self._pe_dev = None
        if use_context:
            # same module tree as the reference's seq2reg ContextFlashAttentionEncoderLayer (seq2reg/modules.py:40-126):
            # mixer.MHA / crossMHA.MHA / norm1-3 / linear_geglu_1-2 / buffer m -- and the same arithmetic as the seq2gene
            # layer of that name (cross attention without ALiBi, q = tokens, k/v = the context rows)
            from ..seq2gene.modules.layers import ContextFlashAttentionEncoderLayer
            self.transformer_encoder = nn.ModuleList(
                [ContextFlashAttentionEncoderLayer(d_model=embedding_dim, nhead=num_heads, batch_first=True,
                                                   use_alibi=use_alibi, mlp_dout=mlp_dout) for _ in range(num_layers)])
            for l in self.transformer_encoder:
                l.mixer.MHA.family, l.crossMHA.MHA.family = "seq2reg_self", "seq2reg_ctx_cross"
        else:
            self.transformer_encoder = nn.ModuleList(
                [FlashTransformerLayer(d_model=embedding_dim, nhead=num_heads, use_alibi=use_alibi) for _ in range(num_layers)])
        if seq_pool == "linear":              # :183-184
            self.linear = nn.Linear(token_length, 1)
        # kept only so that checkpoints load strictly; unused at inference
        in_f = embedding_dim * 2 if strand_agg == "concat" else embedding_dim
        self.tissue_classifiers = nn.ModuleDict({str(t): nn.Linear(in_f, num_classes) for t in range(num_tissues)})

    def _pos_table(self, device):
        if self.pos_encoding_type != "sinusoidal":
            return None
        if self._pe_dev is None or self._pe_dev.device != device:
            self._pe_dev = self.position_encoding.to(device).contiguous()
        return self._pe_dev

    def _layer0_qkv_table(self, device):
        """16-bit [vocab * key_L, 3 d]: the first layer's packed_qkv_ln(norm1) projection of EVERY possible encoder input row
        x0 = Embedding(id) + positional(position) (key = id * key_L + position; key_L = token_length, or 1 with ALiBi).  The
        rows are made by the same kernels the tokens would go through (vf_embed_stream on the identity window list, then the
        LayerNorm-consumer GEMM), and a GEMM row depends on nothing but its own input row: looking a token's projection up
        is bit-identical to projecting the token.  Built once per weights / operand type (0.2 ms of GEMM), then every batch
        replaces a [n_tokens, d] x [d, 3 d] GEMM by a row gather."""
        l0 = self.transformer_encoder[0]
        pos = self._pos_table(device)
        V = self.token_embedding.weight.shape[0]
        key_L = pos.shape[0] if pos is not None else 1
        prm = [self.token_embedding.weight, l0.norm1.weight, l0.norm1.bias, l0.MHA.Wqkv.weight, l0.MHA.Wqkv.bias]
        if pos is not None:
            prm.append(pos)                                   # the positional table is an input of every row too
        key = (str(device), key_L) + tuple((p.data_ptr(), p._version) for p in prm)
        slots = self.__dict__.setdefault("_qkv_tabs", {})    # one slot per operand type (a bf16 <-> fp16 switch rebuilds nothing)
        hit = slots.get(ops.cdt())
        if hit is not None and hit[0] == key:
            return hit[1], key_L
        for dt in [d for d, v in slots.items() if v[0] != key]:      # tables of previous weights / another device: 307 MB each
            del slots[dt]                                            # (round-5 advice: they stayed in the other slot forever)
        ids = torch.arange(V, device=device, dtype=torch.int64).view(V, 1).expand(V, key_L).contiguous()
        pad = torch.zeros((V, key_L), dtype=torch.uint8, device=device)
        cu = (torch.arange(V + 1, device=device, dtype=torch.int32) * key_L).contiguous()
        # rows no token may ever show (unused ids) must not raise the LayerNorm-fold alert: the tokens of a batch raise it
        # themselves (their own vf_embed_stream pass), and the recomputation does not come through this table
        alert = ops._alert_flag(device)
        before = alert.clone()
        s = ops.embed_stream(ids, pad, cu, self.token_embedding.weight, pos, V * key_L, need_t16=False)
        alert.copy_(before)
        w, b, c = l0.MHA.packed_qkv_ln(l0.norm1)
        slots[ops.cdt()] = (key, ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16, family="seq2reg"))
        return slots[ops.cdt()][1], key_L

    def _layer0_qkv_table_bytes(self) -> int:
        """Size the table of _layer0_qkv_table would have: vocab * key_L rows (key_L = rows of the positional table, NOT the
        batch's window length) of 3 d 16-bit values."""
        key_L = self.position_encoding.shape[0] if self.pos_encoding_type == "sinusoidal" else 1
        V, d = self.token_embedding.weight.shape
        return int(V) * int(key_L) * 3 * int(d) * 2

    def embed_packed(self, ids: torch.Tensor, pad: torch.Tensor, n_tokens: int, out_dtype=None,
                     max_len: int = 0, context: torch.Tensor | None = None):
        """ids int64 [W, L], pad bool/u8 [W, L] (True = pad) on the GPU -> pooled [W, d].
        n_tokens = number of valid tokens (host-known; sizes the packed buffers); max_len = longest window in valid
        tokens when the host knows it (sizes the attention grid; 0 -> L, an upper bound).
        context int64 [W]: the window's reference-cCRE label, required by a use_context tokenizer (:222-245)."""
        with runtime.forward_env():          # VF_LN_FOLD / VF_TRUNK16 read once (the enclosing model's snapshot, if any)
            return self._embed_packed(ids, pad, n_tokens, out_dtype, max_len, context)

    def _embed_packed(self, ids, pad, n_tokens, out_dtype, max_len, context):
        W, L = ids.shape
        Lmax = max_len if 0 < max_len < L else L
        with ops.scope("seq2reg"):
            cu = ops.mask_to_cu_seqlens(pad)
            from ..seq2gene.modules.layers import ln_fold_enabled, trunk_f16_active
            l0 = self.transformer_encoder[0]
            # (vf_embed_stream serves d <= 2048; wider tokenizers keep the embed_pack + ln_stream form, which has no limit)
            if (trunk_f16_active() and ln_fold_enabled(l0.norm1.weight.numel(), l0.linear_geglu_2.in_features)
                    and self.token_embedding.weight.shape[1] <= 2048):
                # the encoder input as (16-bit copy, fp16 trunk copy, row statistics): its fp32 rows have no reader
                x = ops.embed_stream(ids, pad, cu, self.token_embedding.weight, self._pos_table(ids.device), n_tokens)
            else:
                x = ops.embed_pack(ids, pad, cu, self.token_embedding.weight, self._pos_table(ids.device), n_tokens)
            if self.use_context:
                if context is None or torch.is_floating_point(context):
                    raise NotImplementedError(
                        "a use_context tokenizer needs integer cCRE labels as context; the gene branch of the reference "
                        "passes a float zero tensor (model_combined_modulator.py:575-577), which nn.Embedding rejects there too")
                ctx = self._context_rows(pad, cu, context.to(ids.device).long().reshape(-1), n_tokens)
                from ..seq2gene.modules.layers import trunk16_enabled
                n_layers = len(self.transformer_encoder)
                for li, layer in enumerate(self.transformer_encoder):
                    x = layer.forward_packed(x, cu, Lmax, context=ctx, cu_ctx=cu, max_ctx=Lmax,
                                             keep_x=not trunk16_enabled("seq2reg") or li + 1 == n_layers)
            else:
                from ..seq2gene.modules.layers import trunk16_enabled
                n_layers = len(self.transformer_encoder)
                # mean pool: the last layer returns the pooled rows themselves (its down-projection commutes with the mean)
                lg2 = self.transformer_encoder[-1].linear_geglu_2                 # (vf_segment_mean16 serves widths <= 2048)
                pool_in_layer = (self.seq_pool == "mean" and runtime.switches().pool_before_down_projection and lg2.in_features % 8 == 0 and
                                 lg2.out_features % 8 == 0 and max(lg2.in_features, lg2.out_features) <= 2048)
                qkv0 = None
                V = self.token_embedding.weight.shape[0]
                if (runtime.switches().layer0_qkv_table and ln_fold_enabled(l0.norm1.weight.numel(), l0.linear_geglu_2.in_features) and
                        self.token_embedding.weight.shape[1] <= 2048 and
                        self._layer0_qkv_table_bytes() <= LAYER0_QKV_TABLE_MAX_BYTES):
                    tab, key_L = self._layer0_qkv_table(ids.device)
                    qkv0 = (tab, ops.token_keys(ids, pad, cu, n_tokens, V, key_L))      # gathered by the attention kernel's loads
                for li, layer in enumerate(self.transformer_encoder):
                    x = layer.forward_packed(x, cu, Lmax, last=li + 1 == n_layers, keep_x=not trunk16_enabled("seq2reg"),
                                             pool_mean=pool_in_layer and li + 1 == n_layers, qkv=qkv0 if li == 0 else None)
                if pool_in_layer:
                    od = ops.cdt() if out_dtype is None else out_dtype
                    return x if od == torch.float32 else ops.cast16(x, od)
            if isinstance(x, ops.LnStream):                           # layers exchange (x, bf16 copy, row statistics)
                x = x.x
            if self.seq_pool == "mean":                               # :263-267
                return ops.segment_mean(x, cu, out_dtype)
            if self.seq_pool == "max":                                # :257-261 (pads carry -inf there = valid tokens only)
                m = ops.segment_max(x, cu)
                od = ops.cdt() if out_dtype is None else out_dtype
                return m if od == torch.float32 else ops.cast16(m, od)
            return ops.segment_linear(x, cu, pad, self.linear.weight.reshape(-1).contiguous(), self.linear.bias, out_dtype)

    def _context_rows(self, pad, cu, labels, n_tokens):
        """fp32 [n_tokens, d]: context_embedding(label of the token's window), optionally expanded per position
        (expand_context = nn.Linear(1, token_length) applied to each embedding element, :229-236:
        context[p, :] = emb * W[p] + b[p]).  Index bookkeeping by torch, row arithmetic by vf_affine_rows_f32."""
        lens = (cu[1:] - cu[:-1]).long()
        tok_label = torch.repeat_interleave(labels, lens, output_size=n_tokens).contiguous()
        table = self.context_embedding.weight.float().contiguous()
        if not self.expand_context_type:
            return ops.gather_rows_f32(table, None, tok_label)
        pos = torch.nonzero(pad.view(torch.uint8) == 0)[:, 1]                                   # position of every packed token
        scale = self.expand_context.weight[:, 0].float()[pos].contiguous()
        shift = self.expand_context.bias.float()[pos].contiguous()
        return ops.affine_rows(table, tok_label, scale, shift)

    def forward(self, x, padding_mask, tissue_vector=None, context=None, only_embed=False, precision=torch.float32):
        """x int64 [b, strands, L]; padding_mask bool [b, strands, L] (True = pad); context int64 [b] (use_context)
        -> fp32 [b, strands, d] (reference seq2reg/model.py:193-279 with only_embed=True)."""
        if not only_embed:
            raise NotImplementedError("only the embedding path (only_embed=True) is part of the inference hot path")
        b, ns, L = x.size()
        dev = self.token_embedding.weight.device
        ids = x.reshape(b * ns, L).to(dev).long().contiguous()
        pad = padding_mask.reshape(b * ns, L).to(dev).contiguous()
        n_tokens = int((~pad.bool()).sum().item())
        ctx = None
        if self.use_context and context is not None:
            ctx = torch.as_tensor(context).to(dev).reshape(-1)
            ctx = ctx if torch.is_floating_point(ctx) else ctx.long().repeat_interleave(ns)
        return self.embed_packed(ids, pad, n_tokens, torch.float32, context=ctx).view(b, ns, -1)
