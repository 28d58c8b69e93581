"""seq2reg transformer layer on the HIP kernels (reference: seq2reg/modules.py:129-191)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..seq2gene.modules.layers import MHA, get_alibi_slopes, packed_linear  # noqa: F401


class FlashTransformerLayer(nn.Module):
    """LN1 -> self-MHA (ALiBi iff configured) -> +src -> LN2 -> GeGLU(d -> 2048 -> 1024 -> d) -> +src
    (seq2reg/modules.py:149-191; both residuals add the layer INPUT, :152-153,179,188).

    The reference re-pads after attention and zeroes the padded rows (:171-178); pads never feed a valid
    token (attention is varlen, the rest is per-token) and are dropped by the pooling mask, so this
    implementation keeps the stream packed from the embedding to the pool."""

    def __init__(self, d_model, nhead, hidden_dim=2048, dropout=0.1, use_alibi=False, mlp_dout=0.1):
        super().__init__()
        self.MHA = MHA(d_model, nhead, dropout=dropout, use_flash_attn=True, use_alibi=use_alibi)
        self.MHA.family = "seq2reg_self"
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear_geglu_1 = nn.Linear(d_model, hidden_dim)
        self.dropout = nn.Dropout(mlp_dout)
        self.linear_geglu_2 = nn.Linear(hidden_dim // 2, d_model)

    def _pooled_down_projection(self, hg, w2, b2, cu, res_f32=None, res16=None, res16_scale: float = 1.0):
        """mean over each window's tokens of (src + linear_geglu_2(hg)) evaluated as mean(src) + W2 mean(hg) + b: the pool
        (seq2reg/model.py:263-267) is linear and so is the layer's last operation (seq2reg/modules.py:184-188), so the
        down-projection runs on W pooled rows instead of n_tok token rows and the fp32 token rows of the encoder output are
        never written.  mean(hg) stays fp32-accurate through the 16-bit GEMM as [hi | lo] against [W2 | W2]; the residual
        mean is taken over the very values the token-level epilogue adds: the fp32 rows, or a 16-bit copy times its scale
        (res16: the fp16 trunk copy, or the operand-type copy of the layer input).  -> fp32 [W, d]."""
        wm = self.linear_geglu_2.weight                  # keyed on the MASTER parameter (as packed_linear is): the packed copy's
        key = (wm.data_ptr(), wm._version, str(wm.device), ops.cdt())      # address can be reused by its successor
        if getattr(self, "_w2_split_key", None) != key:
            self._w2_split, self._w2_split_key = torch.cat([w2, w2], dim=1).contiguous(), key
        ph = ops.segment_mean16(hg, cu, split=True)
        pr = ops.segment_mean16(res16, cu, in_scale=res16_scale) if res16 is not None else \
            ops.segment_mean(res_f32, cu, torch.float32)
        return ops.gemm(ph, self._w2_split, b2, ops.EPI_RES_F32, residual=pr, family="seq2reg")

    def forward_packed(self, src, cu: torch.Tensor, max_seqlen: int, last: bool = False, keep_x: bool = True,
                       pool_mean: bool = False, qkv: tuple | None = None):
        """src: fp32 [tokens, d] or an ops.LnStream; returns the same kind (an LnStream when LayerNorm is folded into
        the GEMMs, see seq2gene.modules.layers.ln_fold_enabled; a plain tensor from the `last` layer).  keep_x=False: the
        result's fp32 rows have no reader (16-bit trunk, layers.trunk16_enabled) and are not stored.
        pool_mean (with last): return the per-window MEAN of the layer's output rows, fp32 [W, d], instead of the rows
        (_pooled_down_projection).  qkv (folded path): this layer's packed_qkv_ln projection of norm1(src), already made
        (the encoder's first layer looks it up per distinct input row: (table, row per token) of Seq2RegPredictor._layer0_qkv_table)."""
        from ..seq2gene.modules.layers import (_as_stream, _as_tensor, down_projection, ln_fold_enabled, packed_linear_ln,
                                               trunk_f16_active)
        if ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            s = _as_stream(src)
            a = self.MHA.attend_ln(s, self.norm1, None, cu, max_seqlen, None, None) if qkv is None else \
                self.MHA.attend_qkv(qkv[0], cu, max_seqlen, rows=qkv[1])
            # x1 is read only through norm2 -> linear_geglu_1: no fp32 store, and its residual is the 16-bit copy of the input
            x1 = self.MHA.out_ln(a, s, need_x=False)
            w1, b1, c1 = packed_linear_ln(self.linear_geglu_1, self.norm2, geglu=True)
            hg = ops.gemm_ln_consumer(x1, w1, b1, c1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            if last and pool_mean:          # the same residual operand down_projection would add per token
                if trunk_f16_active():
                    return self._pooled_down_projection(hg, w2, b2, cu, res16=s.t16 if s.t16 is not None else ops.trunk16_of(s.x),
                                                        res16_scale=1.0 / ops.T16_SCALE)
                if s.x is None:
                    return self._pooled_down_projection(hg, w2, b2, cu, res16=s.x16, res16_scale=1.0 / s.scale)
                return self._pooled_down_projection(hg, w2, b2, cu, res_f32=s.x)
            if last:        # the encoder's last layer feeds the pooling, not a LayerNorm: plain fp32 result
                if s.x is None or trunk_f16_active():    # 16-bit trunk: the layer input exists as its trunk copy only
                    return down_projection(hg, w2, b2, s, keep_x=True, need_t16=False).x
                return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=s.x)
            # 16-bit trunk (layers.trunk16_enabled): the next layer reads the 16-bit copies + statistics only
            return down_projection(hg, w2, b2, s, keep_x)
        src = _as_tensor(src)
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
        x1 = self.MHA.fused(h, src, cu, max_seqlen)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        if last and pool_mean:
            return self._pooled_down_projection(hg, w2, b2, cu, res_f32=src)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)

    def forward(self, src, src_key_padding_mask=None, precision=torch.float32):
        """Reference signature on padded [b, L, d] input; padded rows of the result are left as the
        reference leaves them only where they matter (valid rows); pad rows are returned as zeros."""
        from ..seq2gene.modules.layers import _as_tensor, pad_input, unpad_input, _cu_from_padded
        b, L = src.shape[:2]
        if src_key_padding_mask is None:
            out = _as_tensor(self.forward_packed(src.reshape(b * L, -1).float().contiguous(), _cu_from_padded(b, L, src.device), L))
            return out.view(b, L, -1).to(src.dtype)
        xs, idx, cu, mx, _ = unpad_input(src, ~src_key_padding_mask)
        return pad_input(_as_tensor(self.forward_packed(xs, cu, mx)), idx, b, L).to(src.dtype)
