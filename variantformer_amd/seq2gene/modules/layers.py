"""seq2gene layers on the MI355X HIP kernels, with the reference's class names, constructor arguments
and state-dict keys (reference: seq2gene/modules/layers.py; flash_attn.modules.mha.MHA [3p]).

Data layout: every stream is PACKED -- [total_tokens, D] with int32 cu_seqlens -- for the whole stack
(the reference unpads once and threads cu_seqlens through, model_combined_modulator.py:170-233).
The residual stream is fp32 in HBM; GEMM / attention operands are bf16 (DESIGN.md "Rounding points").
Padded [B,S,D] inputs with boolean masks are accepted at the same call sites the reference accepts
them and are packed on entry.

Nothing here falls back to PyTorch math: each step is a call into libvf_hip.so (variantformer_amd.ops).
"""
from __future__ import annotations

import math
import threading
from typing import Optional

import torch
import torch.nn as nn

from ... import ops, runtime


def get_alibi_slopes(n: int) -> torch.Tensor:
    """ALiBi slopes; same values as the reference helper (layers.py:15-37) and flash-attn's."""
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]

    def slopes(n):
        if math.log2(n).is_integer():
            return pow2(n)
        c = 2 ** math.floor(math.log2(n))
        return pow2(c) + slopes(2 * c)[0::2][: n - c]
    return torch.tensor(slopes(n))


# ---------------------------------------------------------------------------------------------
# weight packing (fp32 master parameters -> bf16 kernel operands), cached per parameter version
# ---------------------------------------------------------------------------------------------
def _row_scale(n_rows: int, scale: float, qrows: int, device):
    """[n_rows, 1] factor: `scale` for the first `qrows` output rows (all rows when qrows = 0), 1 elsewhere."""
    v = torch.full((n_rows, 1), float(scale), dtype=torch.float32, device=device)
    if qrows:
        v[qrows:] = 1.0
    return v


def packed_linear(lin: nn.Linear, geglu: bool = False, wscale: float = 1.0, bscale: float = 1.0, qrows: int = 0):
    """(16-bit weight [N,K] in the current compute dtype, fp32 bias [N]) for vf_gemm_*; `geglu` applies the GEGLU row
    interleave.  Rebuilt whenever the parameter is modified in place (load_state_dict), moved, or the compute dtype
    changes (one cached copy per module: a model runs in one precision at a time).  `wscale` (a power of two): the
    16-bit weights are 16bit(W * wscale) -- the operand of a GEMM whose OTHER operand is a stream copy stored scaled by
    1 / wscale (ops.LnStream.scale, fp16 mode); the product is then the unscaled one, exactly.  `bscale` multiplies the
    bias (q_prescale passes wscale = bscale = softmax scale * log2 e: the whole projection is scaled).  `qrows` > 0: both
    scales apply to the first `qrows` output rows only (the Q rows of a packed Wqkv)."""
    w = lin.weight
    key = (w.data_ptr(), w._version, str(w.device), geglu, ops.cdt(), float(wscale), float(bscale), int(qrows),
           None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version))
    cache = getattr(lin, "_vf_packed", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    with torch.no_grad():
        wf = w.detach().float()
        wb = ops.cast16((wf if wscale == 1.0 else wf * _row_scale(wf.shape[0], wscale, qrows, wf.device)).contiguous())
        b = None
        if lin.bias is not None:
            b = lin.bias.detach().float()
            if bscale != 1.0:
                b = b * _row_scale(b.shape[0], bscale, qrows, b.device)[:, 0]
            b = b.contiguous()
        if geglu:
            wb, b = ops.pack_geglu_rows(wb, b)
    lin._vf_packed = (key, wb, b)
    lin._vf_packed_ln = None          # one operand copy per Linear: a module runs folded or unfolded, not both at once
    return wb, b


def packed_linear_ln(lin: nn.Linear, norm: nn.LayerNorm, geglu: bool = False, wscale: float = 1.0, qrows: int = 0):
    """Operands of a GEMM that consumes LayerNorm(x) without a normalised copy of x (ops.gemm_ln_consumer):
         w'     = bf16(gamma (.) W)        [N, K]   (gamma scales the K columns)
         bias'  = W . beta + b             [N] fp32 (from the fp32 master weights)
         colsum = sum_k float(w'[n, k])    [N] fp32
       so that LN(x) W^T + b = rstd * (x w'^T - mean * colsum) + bias'.  `geglu` applies the GEGLU row interleave to all
       three.  `wscale`: the projection (its first `qrows` output rows; all of them when qrows = 0) multiplied by a constant
       (q_prescale: the softmax scale folded into Wq / the Q rows of Wqkv): w' = 16bit(wscale * gamma (.) W), bias' =
       wscale * (W . beta + b).  One-time weight
       preparation (cached per parameter versions), torch elementwise / reduce ops on the fp32 masters; nothing of this
       runs per batch."""
    w, g, be = lin.weight, norm.weight, norm.bias
    key = (w.data_ptr(), w._version, g.data_ptr(), g._version, be.data_ptr(), be._version, str(w.device), geglu, ops.cdt(),
           float(wscale), int(qrows), None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version))
    cache = getattr(lin, "_vf_packed_ln", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2], cache[3]
    with torch.no_grad():
        wf = w.detach().float()
        wg = wf * g.detach().float()[None, :]
        rs = None if wscale == 1.0 else _row_scale(wf.shape[0], wscale, qrows, wf.device)   # first `qrows` rows (0: all)
        wb = ops.cast16((wg if rs is None else wg * rs).contiguous())                # current operand type (bf16 / fp16)
        b = wf @ be.detach().float()
        if lin.bias is not None:
            b = b + lin.bias.detach().float()
        if rs is not None:
            b = b * rs[:, 0]
        b = b.contiguous()
        if geglu:
            wb, b = ops.pack_geglu_rows(wb, b)
        colsum = wb.float().sum(dim=1).contiguous()
    lin._vf_packed_ln = (key, wb, b, colsum)
    lin._vf_packed = None             # drop the plain 16-bit copy a previous unfolded / fp16 run may have left
    return wb, b, colsum


_LN_FOLD_TLS = threading.local()       # .off > 0: this THREAD is inside ln_fold_forced_off() (a recomputation in one thread must
                                       # not flip the path of a forward running in another)
_LN_FOLD_DISABLED = False              # ln_fold_disable(): off for the whole process, on purpose


class ln_fold_forced_off:
    """Context: every layer (of forwards issued by THIS thread) takes the separate-LayerNorm path on fp32 rows, exactly as with
    VF_LN_FOLD=0, whatever the environment says.  The model recomputes a batch under it when the statistics kernels flagged
    rows the folded form does not serve (ops.ln_fold_alert_take): the same kernels, the same arithmetic, the same bits as a
    VF_LN_FOLD=0 run."""

    def __enter__(self):
        _LN_FOLD_TLS.off = getattr(_LN_FOLD_TLS, "off", 0) + 1
        return self

    def __exit__(self, *exc):
        _LN_FOLD_TLS.off -= 1
        return False


def ln_fold_disable():
    """Switch the fold off for the rest of the process (a checkpoint that keeps tripping the alert: stop paying for two
    forwards and two weight packings per batch)."""
    global _LN_FOLD_DISABLED
    _LN_FOLD_DISABLED = True


def ln_fold_reenable():
    """Undo ln_fold_disable() (tests; a driver that swapped checkpoints)."""
    global _LN_FOLD_DISABLED
    _LN_FOLD_DISABLED = False


def ln_fold_enabled(*widths: int) -> bool:
    """LayerNorm folded into the neighbouring GEMMs when EVERY contraction width of the layer's
    folded GEMMs (d_model, and hidden_dim / 2 for the down-projection producer) is one the MFMA path takes (K % 64 == 0);
    otherwise -- and with VF_LN_FOLD=0, inside ln_fold_forced_off(), or for a model whose self-healing switched the fold off
    -- the separate LayerNorm pass.  VF_LN_FOLD is read once per forward (runtime.forward_env).  Both operand types: an fp16
    stream's 16-bit copy is stored scaled by a power of two (ops.x16_scale_for), so the raw residual cannot leave the fp16
    range."""
    return (not _LN_FOLD_DISABLED and getattr(_LN_FOLD_TLS, "off", 0) == 0 and all(int(w) % 64 == 0 for w in widths)
            and runtime.env().ln_fold)


def trunk16_enabled(stack: str = "modulator") -> bool:
    """A layer's OUTPUT (the trunk: W2.h + layer input) travels to the next layer of its stack as 16-bit copies + row
    statistics only, no fp32 rows in either direction; the LAST layer of a stack (its output is pooled / returned, not fed
    to a LayerNorm -> Linear pair) keeps its fp32 result.  Part of the LayerNorm fold.  VF_TRUNK16 selects what the next
    down-projection adds as its residual (trunk16_mode):
      f16 (default)  a scaled FP16 copy of the trunk written beside the operand-type copy (down_projection,
                     vf_gemm_ln_t16): 6 instead of 10 bytes per element through the epilogue, 11 significant bits -- the
                     embeddings' distance from pure fp32 arithmetic stays at the fp32 trunk's (3.3e-3 vs 3.4e-3 on a
                     full-depth gene), step +1 %; with fp16 operands the operand copy already is that trunk;
      0              fp32 rows.
    (Round 3 also measured the reference's own rounding point -- the trunk in the OPERAND type, bf16 -- as modes "1" / "s2r":
    -3.2 % / -1.2 % step time, but expression error 3.4e-4 -> 8.8e-4 against a bar of 1e-3; removed in round 4, DESIGN.md
    appendix.)  oracle.Rounding(trunk16=...) restates the rounding points."""
    return trunk16_mode() == "f16"


def trunk16_mode() -> str:
    """VF_TRUNK16 (read once per forward, runtime.forward_env): "f16" (default) -- the trunk travels between the layers of
    a stack as a scaled FP16 copy whatever the operand type (down_projection below; with fp16 operands that is the
    operand-type copy itself); "0" -- fp32 rows."""
    return runtime.env().trunk16


def trunk_f16_active() -> bool:
    """The fp16 trunk copy is in use: mode "f16" with bf16 operands (an fp16 operand copy already IS the fp16 trunk:
    trunk16_enabled's plain 16-bit path serves it)."""
    return trunk16_mode() == "f16" and ops.cdt() == torch.bfloat16


def down_projection(hg, w2, b2, s, keep_x: bool = True, need_t16: bool | None = None):
    """The layer's output stream x = linear_geglu_2(hg) + (layer input s) with its 16-bit copy and row statistics
    (LayerNorm fold).  The residual is, in this order: the fp16 trunk copy of the layer input (trunk_f16_active: taken
    from the stream, or made from its fp32 rows for the first layer of a stack -- EVERY layer of a folded stack then adds
    fp16(x_in * 2^-4) * 2^4, oracle.Rounding.trunk), its fp32 rows, or its operand-type copy (trunk16 modes "s2r" / "1").
    keep_x: store the fp32 rows (the last layer of a stack); otherwise the next layer reads the copies only."""
    if trunk_f16_active():
        t = s.t16 if s.t16 is not None else ops.trunk16_of(s.x)
        return ops.gemm_ln_producer(hg, w2, b2, None, need_x=keep_x, trunk16=t,
                                    need_t16=(not keep_x) if need_t16 is None else need_t16)
    return ops.gemm_ln_producer(hg, w2, b2, _ffn_residual(s), need_x=keep_x)


def _ffn_residual(s):
    """The layer input as the residual of the down-projection: its fp32 rows when they exist (the first layer of a stack,
    or VF_TRUNK16=0), else the stream itself = its 16-bit copy."""
    return s.x if s.x is not None else s


def _as_stream(x):
    return x if isinstance(x, ops.LnStream) else ops.ln_stream(x)


def _as_tensor(x):
    return x.x if isinstance(x, ops.LnStream) else x


def _cu_from_padded(batch: int, seqlen: int, device) -> torch.Tensor:
    return torch.arange(0, batch + 1, dtype=torch.int32, device=device) * seqlen


def unpad_input(hidden: torch.Tensor, keep_mask: torch.Tensor):
    """flash_attn.bert_padding.unpad_input [3p] contract: (packed, indices int64, cu_seqlens int32,
    max_seqlen int, seqlens).  Index bookkeeping by torch; the row move is vf_gather_rows_f32."""
    seqlens = keep_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(keep_mask.flatten(), as_tuple=False).flatten()
    cu = torch.nn.functional.pad(torch.cumsum(seqlens, 0, dtype=torch.int32), (1, 0))
    flat = hidden.reshape(-1, hidden.shape[-1]).float().contiguous()
    packed = ops.gather_rows_f32(flat, None, indices)
    return packed, indices, cu, int(seqlens.max().item()), seqlens


def pad_input(packed: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    """flash_attn.bert_padding.pad_input [3p]: scatter packed rows into a zero [B,S,D] tensor."""
    inv = torch.full((batch * seqlen,), -1, dtype=torch.int64, device=packed.device)
    inv[indices] = torch.arange(indices.numel(), device=packed.device)
    zero = torch.zeros((1, packed.shape[-1]), dtype=torch.float32, device=packed.device)
    out = ops.gather_rows_f32(packed.float().contiguous(), zero, inv)      # -1 -> row 0 of `zero`
    return out.view(batch, seqlen, packed.shape[-1])


class MHA(nn.Module):
    """Parameter names and call signature of flash_attn.modules.mha.MHA [3p] as the reference uses it
    (layers.py:344-351, 437-439, 465, 482-487; seq2reg/modules.py:140-142,167): Wqkv (self) or Wq+Wkv
    (cross) and out_proj; rows of Wqkv are (three, head, dh), of Wkv (two, head, dh).

    __call__(x[, x_kv], cu_seqlens=, max_seqlen=, cu_seqlens_k=, max_seqlen_k=) on packed [tokens, D]
    (or padded [B,S,D] with no kwargs) returns out_proj(attention) in x's dtype, like the original.
    The layers use `fused()` which also folds the residual add into the out_proj epilogue."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, use_flash_attn=True, use_alibi=False, cross_attn=False, **kw):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.cross_attn = embed_dim, num_heads, cross_attn
        self.head_dim = embed_dim // num_heads
        self.q_log2_scale = math.log2(math.e) / math.sqrt(self.head_dim)     # softmax scale (1 / sqrt(dh)) in base 2
        self.use_alibi = use_alibi
        if cross_attn:
            self.Wq = nn.Linear(embed_dim, embed_dim)
            self.Wkv = nn.Linear(embed_dim, 2 * embed_dim)
        else:
            self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.family = "cross" if cross_attn else "self"      # label of this module's launches in ops.KernelTimer
        if use_alibi:
            self.register_buffer("alibi_slopes", get_alibi_slopes(num_heads).float(), persistent=False)
        else:
            self.alibi_slopes = None

    def packed_qkv(self):
        """packed_linear(Wqkv) with the Q rows carrying the base-2 softmax scale 1 / sqrt(dh) * log2(e): q . k leaves the
        matrix pipe as the base-2 logit (ops.attn_varlen(q_log2=True)) -- one rounding of the scaled weights instead of one of the
        unscaled ones; oracle.Rounding(q_prescale=...) restates the rounding point.  What it buys: the long-stream attention
        kernel drops the running maximum and the multiply-add in front of every exponential (gene -> CRE cross attention -14 %)."""
        c = self.q_log2_scale
        return packed_linear(self.Wqkv, wscale=c, bscale=c, qrows=self.embed_dim)

    def packed_qkv_ln(self, norm: nn.LayerNorm):
        """packed_linear_ln(Wqkv, norm), Q rows as in packed_qkv."""
        return packed_linear_ln(self.Wqkv, norm, wscale=self.q_log2_scale, qrows=self.embed_dim)

    # -- pieces -------------------------------------------------------------------------------
    def project_kv(self, x_kv_bf16: torch.Tensor) -> torch.Tensor:
        """bf16 [tokens_k, 2D] = Wkv(x_kv): exposed so that a caller can compute it once and share it."""
        w, b = packed_linear(self.Wkv)
        return ops.gemm(x_kv_bf16, w, b, ops.EPI_BF16)

    def project_kv_of(self, ctx) -> torch.Tensor:
        """project_kv of a context stream: an fp32 tensor (cast here), or an ops.LnStream, whose 16-bit copy is the
        operand -- un-normalised use, so a SCALED copy (fp16 mode) meets weights scaled by the inverse power of two:
        (x * s) . (W / s) = x . W in the fp32 accumulator, with the mantissas of fp16(x) and fp16(W)."""
        if not isinstance(ctx, ops.LnStream):
            return self.project_kv(ops.cast16(ctx))
        if ctx.scale == 1.0:
            return self.project_kv(ctx.x16)
        w, b = packed_linear(self.Wkv, wscale=1.0 / ctx.scale)
        return ops.gemm(ctx.x16, w, b, ops.EPI_BF16)

    def attend(self, x_bf16, kv_bf16, cu_q, max_q, cu_k, max_k) -> torch.Tensor:
        """bf16 attention output [tokens_q, D] (before out_proj)."""
        D = self.embed_dim
        if self.cross_attn:
            c = self.q_log2_scale                                    # q carries the base-2 softmax scale
            w, b = packed_linear(self.Wq, wscale=c, bscale=c)
            q = ops.gemm(x_bf16, w, b, ops.EPI_BF16)
            return ops.attn_varlen(q, kv_bf16[:, :D], kv_bf16[:, D:], cu_q, cu_k, max_q, max_k, self.num_heads,
                                   self.head_dim, self.alibi_slopes, family=self.family, q_log2=True)
        w, b = self.packed_qkv()
        qkv = ops.gemm(x_bf16, w, b, ops.EPI_BF16)
        return self.attend_qkv(qkv, cu_q, max_q)

    def attend_ln(self, s: "ops.LnStream", norm: nn.LayerNorm, kv_bf16, cu_q, max_q, cu_k, max_k) -> torch.Tensor:
        """attend(LayerNorm(s.x)) with the LayerNorm folded into the Wqkv / Wq projection."""
        D = self.embed_dim
        if self.cross_attn:
            w, b, c = packed_linear_ln(self.Wq, norm, wscale=self.q_log2_scale)
            q = ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16)
            return ops.attn_varlen(q, kv_bf16[:, :D], kv_bf16[:, D:], cu_q, cu_k, max_q, max_k, self.num_heads,
                                   self.head_dim, self.alibi_slopes, family=self.family, q_log2=True)
        w, b, c = self.packed_qkv_ln(norm)
        qkv = ops.gemm_ln_consumer(s, w, b, c, ops.EPI_BF16)
        return self.attend_qkv(qkv, cu_q, max_q)

    def lowrank_tables(self, norm, table: torch.Tensor):
        """Weights of the LOW-RANK form of a cross attention whose keys / values are Wkv of the C rows of `table` (fp32 [C, D]):
            logits[t, h, c] = LN(x_t) . z[h, c] + zb[h, c],   z[h, c] = scale * Wq_h^T k[c, h],  zb = scale * bq_h . k[c, h]
            out_proj(sum_c w_c v[c, h])  =  w . U^T + bo,     U[:, h, c] = Wo[:, h] v[c, h]
        with k, v = Wkv table + bkv in fp32 (no 16-bit rounding of q, k, v themselves: each product matrix is rounded once).
        Returns (wz 16-bit [H * Cp, D], bz fp32, colsum fp32 | None, u 16-bit [D, H * Cp], bo fp32, Cp) -- wz carries the
        LayerNorm fold of `norm` (gamma folded in, W beta in the bias, colsum for the mean correction) when norm is given --
        or None when no slot count Cp in [C, 16] makes H * Cp a multiple of 64.  Cached per (weights, operand type)."""
        assert self.cross_attn
        C, D, H, dh = table.shape[0], self.embed_dim, self.num_heads, self.head_dim
        Cp = next((c for c in range(C + (C & 1), 17, 2) if (H * c) % 64 == 0), None)
        if Cp is None:
            return None
        prm = [table, self.Wq.weight, self.Wq.bias, self.Wkv.weight, self.Wkv.bias, self.out_proj.weight, self.out_proj.bias]
        if norm is not None:
            prm += [norm.weight, norm.bias]
        key = (ops.cdt(), norm is not None) + tuple((p.data_ptr(), p._version) for p in prm)
        slot = "_vf_lowrank_ln" if norm is not None else "_vf_lowrank"
        c = getattr(self, slot, None)
        if c is not None and c[0] == key:
            return c[1]
        with torch.no_grad():
            kv = table.detach().float() @ self.Wkv.weight.detach().float().t() + self.Wkv.bias.detach().float()     # [C, 2D]
            k, v = kv[:, :D].view(C, H, dh), kv[:, D:].view(C, H, dh)
            wq = self.Wq.weight.detach().float().view(H, dh, D)
            z = torch.zeros((H, Cp, D), dtype=torch.float32, device=table.device)
            z[:, :C] = torch.einsum("chd,hdk->hck", k, wq) * self.q_log2_scale
            zb = torch.zeros((H, Cp), dtype=torch.float32, device=table.device)
            zb[:, :C] = torch.einsum("chd,hd->hc", k, self.Wq.bias.detach().float().view(H, dh)) * self.q_log2_scale
            z, zb = z.view(H * Cp, D), zb.view(H * Cp)
            u = torch.zeros((D, H, Cp), dtype=torch.float32, device=table.device)
            u[:, :, :C] = torch.einsum("nhd,chd->nhc", self.out_proj.weight.detach().float().view(D, H, dh), v)
            u16 = ops.cast16(u.view(D, H * Cp).contiguous())
            bo = self.out_proj.bias.detach().float().contiguous()
            if norm is not None:
                wz = ops.cast16((z * norm.weight.detach().float()[None, :]).contiguous())
                bz = (z @ norm.bias.detach().float() + zb).contiguous()
                cs = wz.float().sum(dim=1).contiguous()
            else:
                wz, bz, cs = ops.cast16(z.contiguous()), zb.contiguous(), None
        out = (wz, bz, cs, u16, bo, Cp)
        setattr(self, slot, (key, out))
        return out

    def cross_lowrank(self, x, norm, table, log2_count, cu_q, max_q, residual, tables=None):
        """out_proj(cross attention of LayerNorm(x) against Wkv(table rows), row c counted 2^log2_count[s, c] times) + residual
        in the low-rank form (lowrank_tables): two skinny GEMMs around vf_softmax_counted.  x: an ops.LnStream (norm folded)
        -> returns an LnStream (16-bit copy + statistics, no fp32 rows), or, norm=None, an already normalised 16-bit tensor with
        an fp32 residual -> fp32 rows."""
        wz, bz, cs, u16, bo, Cp = tables if tables is not None else self.lowrank_tables(norm, table)
        if norm is not None:
            sc = ops.gemm_ln_consumer(x, wz, bz, cs, ops.EPI_F32)
        else:
            sc = ops.gemm(x, wz, bz, ops.EPI_F32)
        w16 = ops.softmax_counted(sc, log2_count, cu_q, max_q, self.num_heads, Cp, family=self.family)
        if norm is not None:
            return ops.gemm_ln_producer(w16, u16, bo, residual, need_x=False)
        return ops.gemm(w16, u16, bo, ops.EPI_RES_F32, residual=residual)

    def attend_counted(self, x, norm, counted, cu_q, max_q) -> torch.Tensor:
        """Cross attention of LayerNorm(x) (x: an ops.LnStream with `norm` folded into Wq, or, norm=None, an already
        normalised 16-bit tensor) against keys that are copies of a few distinct rows: counted = (kv_table 16-bit [C, 2D],
        log2_count fp32 [n_seq, C]) -- ops.attn_counted_keys."""
        assert self.cross_attn and self.alibi_slopes is None
        c = self.q_log2_scale                                    # q always carries the base-2 softmax scale here
        if norm is not None:
            w, b, cs = packed_linear_ln(self.Wq, norm, wscale=c)
            q = ops.gemm_ln_consumer(x, w, b, cs, ops.EPI_BF16)
        else:
            w, b = packed_linear(self.Wq, wscale=c, bscale=c)
            q = ops.gemm(x, w, b, ops.EPI_BF16)
        return ops.attn_counted_keys(q, counted[0], counted[1], cu_q, max_q, self.num_heads, self.head_dim, family=self.family)

    def attend_qkv(self, qkv, cu_q, max_q, rows=None) -> torch.Tensor:
        """self attention on a packed [tokens, 3D] 16-bit projection (rows ordered (three, head, dh)) made with packed_qkv /
        packed_qkv_ln (its Q third carries the base-2 softmax scale).
        rows int64 [tokens]: qkv is a TABLE of distinct projected rows and token t's row is rows[t] (the first layers'
        projection by lookup): the attention kernel gathers in its loads where it can (ops.attn_rows_supported); otherwise
        the rows are gathered first -- the same bits either way."""
        D = self.embed_dim
        if rows is not None and not (runtime.switches().rows_in_attention and ops.attn_rows_supported(
                self.head_dim, self.alibi_slopes is not None, cu_q.numel() - 1, self.num_heads, max_q, max_q, True)):
            qkv, rows = ops.gather_rows_bf16(qkv, rows), None
        return ops.attn_varlen(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], cu_q, None, max_q, max_q,
                               self.num_heads, self.head_dim, self.alibi_slopes, family=self.family,
                               q_log2=True, rows=rows)

    def out_ln(self, a_bf16, residual_f32, need_x: bool = True) -> "ops.LnStream":
        """out_proj(a) + residual as an LnStream (fp32 stream, its 16-bit copy, row statistics for the next LayerNorm).
        `residual_f32`: the fp32 rows, or an LnStream whose 16-bit copy is the residual (the attention-block sums
        x1 = out_proj(attn) + src, x2 = out_proj(cross) + x1 are read only through the next LayerNorm -> Linear pair and as
        the next block's residual, both in 16 bits: DESIGN.md section 3 item 9; oracle.Rounding(res16=True)).
        need_x=False when the sum is only ever read through the next LayerNorm -> Linear pair (the layers add their FFN
        to the layer INPUT, reference layers.py:99,163 / seq2reg/modules.py:188, so the stream after the last attention
        block has no other reader): the fp32 values are then not written to HBM at all."""
        w, b = packed_linear(self.out_proj)
        return ops.gemm_ln_producer(a_bf16, w, b, residual_f32, need_x=need_x)

    def fused(self, x_bf16, residual_f32, cu_q, max_q, kv_bf16=None, cu_k=None, max_k=None) -> torch.Tensor:
        """fp32 [tokens, D] = out_proj(attention(x)) + residual (one GEMM epilogue)."""
        a = self.attend(x_bf16, kv_bf16, cu_q, max_q, cu_k, max_k)
        w, b = packed_linear(self.out_proj)
        return ops.gemm(a, w, b, ops.EPI_RES_F32, residual=residual_f32)

    # -- flash_attn-compatible call -------------------------------------------------------------
    def forward(self, x, x_kv=None, cu_seqlens=None, max_seqlen=None, cu_seqlens_k=None, max_seqlen_k=None, **kw):
        in_dtype, in_shape = x.dtype, x.shape
        if cu_seqlens is None:                      # padded [B,S,D], no masking (layers.py:487)
            B, S = x.shape[:2]
            cu_seqlens, max_seqlen = _cu_from_padded(B, S, x.device), S
            x = x.reshape(B * S, -1)
            if x_kv is not None:
                Sk = x_kv.shape[1]
                cu_seqlens_k, max_seqlen_k = _cu_from_padded(B, Sk, x.device), Sk
                x_kv = x_kv.reshape(B * Sk, -1)
        cd = x.dtype if x.dtype in (torch.bfloat16, torch.float16) else ops.cdt()     # 16-bit inputs pick the operand type
        with ops.compute_dtype(cd):
            return self._forward_cd(x, x_kv, cu_seqlens, max_seqlen, cu_seqlens_k, max_seqlen_k, in_dtype, in_shape, cd)

    def _forward_cd(self, x, x_kv, cu_seqlens, max_seqlen, cu_seqlens_k, max_seqlen_k, in_dtype, in_shape, cd):
        xb = x if x.dtype == cd else ops.cast16(x.float().contiguous())
        kv = None
        if self.cross_attn:
            src = x_kv if x_kv is not None else x
            kv = self.project_kv(src if src.dtype == cd else ops.cast16(src.float().contiguous()))
            if cu_seqlens_k is None:
                cu_seqlens_k, max_seqlen_k = cu_seqlens, max_seqlen
        a = self.attend(xb, kv, cu_seqlens, max_seqlen, cu_seqlens_k, max_seqlen_k)
        w, b = packed_linear(self.out_proj)
        out = ops.gemm(a, w, b, ops.EPI_F32 if in_dtype == torch.float32 else ops.EPI_BF16)
        return out.to(in_dtype).view(in_shape)


class FlashAttLayer(nn.Module):
    """Varlen wrapper around MHA, self or cross (reference layers.py:328-488)."""

    def __init__(self, d_model, nhead, hidden_dim=2048, dropout=0.1, use_alibi=False, cross_attn=False,
                 flash_attn_3=False):
        super().__init__()
        self.cross_attn = cross_attn
        if cross_attn and flash_attn_3:
            raise NotImplementedError("flash-attention-3 is not supported at this time")
        self.MHA = MHA(d_model, nhead, dropout=dropout, use_flash_attn=True, use_alibi=use_alibi, cross_attn=cross_attn)

    def forward(self, src, cntx=None, src_key_padding_mask=None, context_key_padding_mask=None,
                precision=torch.float32, unpad_info=None, context_unpad_info=None):
        if unpad_info is not None:                                   # already packed (:372-378, :454-467)
            kw = {"cu_seqlens": unpad_info["cu_seqlens"], "max_seqlen": unpad_info["max_seqlen"]}
            if self.cross_attn:
                assert cntx is not None
                if context_unpad_info is not None:
                    kw.update(cu_seqlens_k=context_unpad_info["cu_seqlens"], max_seqlen_k=context_unpad_info["max_seqlen"])
                return self.MHA(src, cntx, **kw)
            return self.MHA(src, **kw)
        if src_key_padding_mask is not None:                         # padded + mask: pack, run, pad back
            batch, seqlen = src.shape[:2]
            xs, idx, cu, mx, _ = unpad_input(src, ~src_key_padding_mask)
            kw = {"cu_seqlens": cu, "max_seqlen": mx}
            if self.cross_attn:
                assert cntx is not None and context_key_padding_mask is not None, \
                    "context_key_padding_mask must be provided if src_key_padding_mask is provided"
                cs, _, cuk, mxk, _ = unpad_input(cntx, ~context_key_padding_mask)
                out = self.MHA(xs, cs, cu_seqlens_k=cuk, max_seqlen_k=mxk, **kw)
            else:
                out = self.MHA(xs, **kw)
            return pad_input(out, idx, batch, seqlen).to(src.dtype)
        return self.MHA(src, cntx) if self.cross_attn else self.MHA(src)


class ContextFlashAttentionEncoderLayer(nn.Module):
    """LN -> self-MHA(ALiBi) -> +src -> LN -> cross-MHA(q = x, kv = context, un-normalised) -> +res ->
    LN -> GeGLU(d -> 2048 -> 1024 -> d) -> + src   (reference layers.py:47-165; note the last residual
    is the LAYER INPUT, :99,163).  Used for the CRE layers and, with only_cross_attention false, for the
    gene layers."""

    def __init__(self, d_model, nhead, hidden_dim=2048, dropout=0.1, batch_first=True, use_alibi=False,
                 make_data_kv=False, mlp_dout=0.0, cross_alibi=False, flash_attn_3=False):
        super().__init__()
        self.mixer = FlashAttLayer(d_model, nhead, dropout=dropout, use_alibi=use_alibi, cross_attn=False)
        self.crossMHA = FlashAttLayer(d_model, nhead, dropout=dropout, use_alibi=cross_alibi, cross_attn=True,
                                      flash_attn_3=flash_attn_3)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.linear_geglu_1 = nn.Linear(d_model, hidden_dim)
        self.dropout = nn.Dropout(mlp_dout)
        self.linear_geglu_2 = nn.Linear(hidden_dim // 2, d_model)
        self.use_alibi = use_alibi
        self.num_heads = nhead
        self.make_data_kv = make_data_kv
        self.activation = nn.GELU()
        if use_alibi:
            self.register_buffer("m", get_alibi_slopes(self.num_heads))

    def self_qkv_of_unique_rows(self, rows_a, rows_b, idx, rows_b_used=None, with_stream: bool = False):
        """LayerNorm1 + Wqkv of a stream whose rows are copies of the rows of two small tables (the gene stream entering
        the FIRST gene layer: every tissue's copy of a gene holds the same chunk rows, only the registry row differs):
        projected once per distinct row; returns (table, row per token).  idx int64 [tokens]: >= 0 row of rows_a, < 0 row -idx-1 of rows_b.
        Exact: LayerNorm and the projection are row-wise.  None when the LayerNorm fold is off (fp16 mode).
        rows_b_used int64 [k] (optional): the rows of table b the stream actually holds -- only THEY may raise the
        LayerNorm-fold alert (an anomalous registry row of a tissue nobody asked for must not flag every batch).
        with_stream: also return the stream itself as the layer consumes it -- an ops.LnStream WITHOUT fp32 rows, its 16-bit
        operand copy (the residual of the self-attention block) and fp16 trunk copy (the residual of the down-projection)
        gathered from the copies of the distinct rows: a cast is row-wise too, so this is bit for bit the copy of the
        gathered fp32 stream, which is then never written or read (round 5: -6 GB per step).  None for that element when the
        layer needs fp32 rows (VF_TRUNK16=0, fp16 operands)."""
        if not ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            return None
        w, b, c = self.mixer.MHA.packed_qkv_ln(self.norm1)
        sa = ops.ln_stream(rows_a.float().contiguous())
        qa = ops.gemm_ln_consumer(sa, w, b, c, ops.EPI_BF16)
        tab_b = rows_b.float().contiguous()
        if rows_b_used is not None:
            ops.ln_stream(tab_b[rows_b_used].contiguous())       # statistics of the rows in use: raises the alert, result unused
        sb = ops.ln_stream(tab_b, raise_alert=rows_b_used is None)
        qb = ops.gemm_ln_consumer(sb, w, b, c, ops.EPI_BF16)
        both = torch.cat([qa, qb], dim=0)                        # (a few thousand rows: index plumbing, not data movement)
        idx2 = torch.where(idx >= 0, idx, rows_a.shape[0] - idx - 1)
        qkv = (both, idx2)           # (table of distinct projected rows, row of every token): gathered by the attention's loads
        if not with_stream:
            return qkv
        stream = None
        if trunk16_enabled() and trunk_f16_active():
            # (bf16 operands with the fp16 trunk copy, the default.  With fp16 operands the FIRST layer of a stack adds its fp32
            # input rows in the down-projection -- oracle.Rounding.trunk -- so that mode keeps the fp32 gather.)
            x16 = ops.gather_rows_bf16(torch.cat([sa.x16, sb.x16], dim=0), idx2)
            t16 = ops.gather_rows_bf16(torch.cat([ops.trunk16_of(sa.x), ops.trunk16_of(sb.x, raise_alert=rows_b_used is None)],
                                                 dim=0), idx2)
            stream = ops.LnStream(None, x16, None, sa.scale, t16)
        return qkv, stream

    def forward_packed(self, src, cu_src, max_src, context=None, cu_ctx=None, max_ctx=None, context_kv=None,
                       cu_cross_q=None, max_cross_q=None, self_qkv=None, keep_x=True, context_counted=None):
        """src fp32 [tokens, D] packed residual stream.  Cross-attention keys/values come either from
        `context` (fp32 packed stream, projected here) or from a precomputed bf16 `context_kv` [tokens_k, 2D].
        `cu_cross_q` lets several self-attention sequences share one K/V block (tissue copies of a gene).
        `self_qkv`: precomputed LayerNorm1 -> Wqkv projection of src (self_qkv_of_unique_rows).
        `keep_x=False` (LayerNorm fold only): the result's fp32 rows have no reader (the next layer of the stack takes
        the 16-bit copy, trunk16_enabled) and are not stored.
        `context_counted` = (kv_table [C, 2D], log2_count [n_seq, C][, table fp32 [C, D]]) instead of context / context_kv: the
        context rows are copies of C distinct rows (the CRE layers' label embeddings) -- MHA.cross_lowrank when the raw table is
        given (two skinny GEMMs around a 9-way softmax), else MHA.attend_counted."""
        if self.make_data_kv:
            return self._forward_packed_data_kv(src, cu_src, max_src, context, cu_ctx, max_ctx)
        cq = cu_src if cu_cross_q is None else cu_cross_q
        mq = max_src if max_cross_q is None else max_cross_q
        if ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            # LayerNorm folded into the GEMMs: every fp32-residual GEMM also emits the bf16 copy + row statistics of its
            # output, every LayerNorm -> Linear pair runs on that copy (no LayerNorm pass, no cast of the context)
            # (the attention-block sums x1, x2 travel as 16-bit copies + statistics, no fp32 rows: section 3 item 9)
            if self_qkv is not None:
                # the projection was computed on the distinct rows; below only the residuals read the stream: its fp32
                # rows / trunk copy (layer output) and its 16-bit copy (self-attention block)
                s = _as_stream(src)
                a = self.mixer.MHA.attend_qkv(self_qkv[0], cu_src, max_src, rows=self_qkv[1])
            else:
                s = _as_stream(src)
                a = self.mixer.MHA.attend_ln(s, self.norm1, None, cu_src, max_src, None, None)
            x1 = self.mixer.MHA.out_ln(a, s, need_x=False)
            lr = None
            if context_counted is not None and runtime.switches().lowrank_context and len(context_counted) > 2:
                lr = self.crossMHA.MHA.lowrank_tables(self.norm2, context_counted[2])
            if lr is not None:
                x2 = self.crossMHA.MHA.cross_lowrank(x1, self.norm2, None, context_counted[1], cq, mq, x1, tables=lr)
            else:
                if context_counted is not None:
                    a = self.crossMHA.MHA.attend_counted(x1, self.norm2, context_counted, cq, mq)
                else:
                    if context_kv is None:
                        context_kv = self.crossMHA.MHA.project_kv_of(context)
                    a = self.crossMHA.MHA.attend_ln(x1, self.norm2, context_kv, cq, mq, cu_ctx, max_ctx)
                x2 = self.crossMHA.MHA.out_ln(a, x1, need_x=False)
            w1, b1, c1 = packed_linear_ln(self.linear_geglu_1, self.norm3, geglu=True)
            hg = ops.gemm_ln_consumer(x2, w1, b1, c1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            return down_projection(hg, w2, b2, s, keep_x)
        src, context = _as_tensor(src), _as_tensor(context)
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
        x1 = self.mixer.MHA.fused(h, src, cu_src, max_src)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        lr = None
        if context_counted is not None and runtime.switches().lowrank_context and len(context_counted) > 2:
            lr = self.crossMHA.MHA.lowrank_tables(None, context_counted[2])
        if lr is not None:
            x2 = self.crossMHA.MHA.cross_lowrank(h, None, None, context_counted[1], cq, mq, x1, tables=lr)
        elif context_counted is not None:
            a = self.crossMHA.MHA.attend_counted(h, None, context_counted, cq, mq)
            w, b = packed_linear(self.crossMHA.MHA.out_proj)
            x2 = ops.gemm(a, w, b, ops.EPI_RES_F32, residual=x1)
        else:
            if context_kv is None:
                context_kv = self.crossMHA.MHA.project_kv(ops.cast16(context))
            x2 = self.crossMHA.MHA.fused(h, x1, cq, mq, context_kv, cu_ctx, max_ctx)
        h = ops.layernorm(x2, self.norm3.weight, self.norm3.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)

    def _forward_packed_data_kv(self, src, cu_src, max_src, context, cu_ctx, max_ctx):
        """make_data_kv=True (reference layers.py:133-136, seq2reg/modules.py:97-100): the roles of the two streams in the
        cross attention are swapped -- its QUERIES are the raw context rows, its keys / values come from norm2(x) -- so its
        output has one row per CONTEXT token, which the reference then adds to the stream (`x += res_short`): context and
        stream must hold the same sequences.  No reference call site enables the option (the models construct their layers
        with the default); it runs on the separate-LayerNorm path (no fold: the normalised stream is a K/V operand here)."""
        src, context = _as_tensor(src), _as_tensor(context)
        if src is None or context is None:
            raise RuntimeError("make_data_kv layers need the fp32 rows of their input streams (run them with VF_LN_FOLD=0 "
                               "when they sit inside a LayerNorm-folded stack)")
        assert context.shape == src.shape and (cu_ctx is None or cu_ctx is cu_src or torch.equal(cu_ctx, cu_src)), \
            "make_data_kv: src and context must have the same shape (reference seq2reg/modules.py:79)"
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
        x1 = self.mixer.MHA.fused(h, src, cu_src, max_src)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        kv = self.crossMHA.MHA.project_kv(h)                               # K / V from the NORMALISED stream
        x2 = self.crossMHA.MHA.fused(ops.cast16(context), x1, cu_src, max_src, kv, cu_src, max_src)     # Q from the raw context
        h = ops.layernorm(x2, self.norm3.weight, self.norm3.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)

    def forward_packed_rows(self, src, cu_src, max_src, rows, cu_rows, context, cu_ctx, max_ctx, cu_cross_rows,
                            max_cross_rows):
        """Same layer, but only the output rows `rows` (int64 indices into src, at most one per self-attention
        sequence and located at position 0 of its sequence) are produced: fp32 [len(rows), D].
        Every token still contributes keys / values to the self attention; the query side, both out-projections, the
        cross attention, LayerNorm 2/3 and the GeGLU FFN run on the selected rows only.  Used for the LAST gene layer,
        whose output is consumed only through the registry token (pool_outputs row 0,
        model_combined_modulator.py:391-392): exact, and ~1/25 of the gene-stream work less."""
        if self.make_data_kv:
            raise NotImplementedError("make_data_kv: the cross attention's rows are context rows; no registry-only form")
        mha = self.mixer.MHA
        if ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            s = _as_stream(src)
            D = s.x16.shape[1]
            w, b, c = mha.packed_qkv_ln(self.norm1)
            kv = ops.gemm_ln_consumer(s, w[D:], b[D:], c[D:], ops.EPI_BF16)            # all rows (K/V need them)
            sr = ops.ln_stream_rows(s, rows)
            q = ops.gemm_ln_consumer(sr, w[:D], b[:D], c[:D], ops.EPI_BF16)            # [R, D]
            a = ops.attn_varlen(q, kv[:, :D], kv[:, D:], cu_rows, cu_src, 1, max_src, mha.num_heads, mha.head_dim,
                                mha.alibi_slopes, q_at_start=True, family=mha.family + "_registry_rows",
                                q_log2=True)
            x1 = mha.out_ln(a, sr, need_x=False)
            ckv = self.crossMHA.MHA.project_kv_of(context)
            a = self.crossMHA.MHA.attend_ln(x1, self.norm2, ckv, cu_cross_rows, max_cross_rows, cu_ctx, max_ctx)
            x2 = self.crossMHA.MHA.out_ln(a, x1, need_x=False)
            w1, b1, c1 = packed_linear_ln(self.linear_geglu_1, self.norm3, geglu=True)
            hg = ops.gemm_ln_consumer(x2, w1, b1, c1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            if sr.x is None or trunk_f16_active():     # 16-bit trunk: the registry rows' residual is their trunk copy
                return down_projection(hg, w2, b2, sr, keep_x=True, need_t16=False).x
            return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=sr.x)
        ctx16 = context.operand16() if isinstance(context, ops.LnStream) else None
        src, context = _as_tensor(src), _as_tensor(context)
        D = src.shape[1]
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)                       # all rows (K/V need them)
        w, b = mha.packed_qkv()
        kv = ops.gemm(h, w[D:], None if b is None else b[D:], ops.EPI_BF16)             # [tokens, 2D]
        hq = ops.gather_rows_bf16(h, rows)
        q = ops.gemm(hq, w[:D], None if b is None else b[:D], ops.EPI_BF16)             # [R, D]
        a = ops.attn_varlen(q, kv[:, :D], kv[:, D:], cu_rows, cu_src, 1, max_src, mha.num_heads, mha.head_dim,
                            mha.alibi_slopes, q_at_start=True, family=mha.family + "_registry_rows",
                            q_log2=True)
        src_rows = ops.gather_rows_f32(src, None, rows)
        wo, bo = packed_linear(mha.out_proj)
        x1 = ops.gemm(a, wo, bo, ops.EPI_RES_F32, residual=src_rows)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        ckv = self.crossMHA.MHA.project_kv(ctx16 if ctx16 is not None else ops.cast16(context))
        x2 = self.crossMHA.MHA.fused(h, x1, cu_cross_rows, max_cross_rows, ckv, cu_ctx, max_ctx)
        h = ops.layernorm(x2, self.norm3.weight, self.norm3.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src_rows)

    def forward(self, src, context, src_key_padding_mask=None, context_padding_mask=None, precision=torch.float32,
                unpad_info=None, context_unpad_info=None, gene_unpad_info=None):
        """Reference signature.  `precision` (a torch dtype or None, as the reference passes it, layers.py:94-125):
        torch.float16 / torch.float32 select fp16 operands (the reference's fp16 flash path), torch.bfloat16 bf16
        operands, None the ambient compute dtype (ops.cdt(), bf16 unless the model's precision says otherwise)."""
        cd = {torch.float16: torch.float16, torch.float32: torch.float16, torch.bfloat16: torch.bfloat16}.get(precision, ops.cdt())
        with ops.compute_dtype(cd):
            return self._forward_ref(src, context, src_key_padding_mask, context_padding_mask, unpad_info,
                                     context_unpad_info, gene_unpad_info)

    def _forward_ref(self, src, context, src_key_padding_mask, context_padding_mask, unpad_info, context_unpad_info,
                     gene_unpad_info):
        if context_padding_mask is None and src_key_padding_mask is not None:
            context_padding_mask = src_key_padding_mask.clone()
        info = gene_unpad_info if gene_unpad_info is not None else unpad_info
        if info is not None:
            cinfo = context_unpad_info if context_unpad_info is not None else info
            return _as_tensor(self.forward_packed(src.float().contiguous(), info["cu_seqlens"], info["max_seqlen"],
                                                  context.float().contiguous(), cinfo["cu_seqlens"], cinfo["max_seqlen"])).to(src.dtype)
        batch, seqlen = src.shape[:2]
        if src_key_padding_mask is None:
            xs, cu, mx, idx = src.reshape(batch * seqlen, -1).float().contiguous(), _cu_from_padded(batch, seqlen, src.device), seqlen, None
            cs, cuk, mxk = context.reshape(-1, context.shape[-1]).float().contiguous(), _cu_from_padded(batch, context.shape[1], src.device), context.shape[1]
        else:
            xs, idx, cu, mx, _ = unpad_input(src, ~src_key_padding_mask)
            cs, _, cuk, mxk, _ = unpad_input(context, ~context_padding_mask)
        out = _as_tensor(self.forward_packed(xs, cu, mx, cs, cuk, mxk))
        if idx is None:
            return out.view(batch, seqlen, -1).to(src.dtype)
        return pad_input(out, idx, batch, seqlen).to(src.dtype)


class FlashAttentionEncoderLayer(nn.Module):
    """CRE layer without a context stream (`use_context: false`): LN -> self-MHA(ALiBi) -> +src -> LN -> GeGLU -> + src
    (reference layers.py:168-228; `norm3` exists there but is never applied, and is kept for the state dict)."""

    def __init__(self, d_model, nhead, hidden_dim=2048, dropout=0.1, batch_first=True, use_alibi=False,
                 make_data_kv=False, mlp_dout=0.0):
        super().__init__()
        self.mixer = FlashAttLayer(d_model, nhead, dropout=dropout, use_alibi=use_alibi, cross_attn=False)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.linear_geglu_1 = nn.Linear(d_model, hidden_dim)
        self.dropout = nn.Dropout(mlp_dout)
        self.linear_geglu_2 = nn.Linear(hidden_dim // 2, d_model)
        self.use_alibi, self.num_heads, self.make_data_kv = use_alibi, nhead, make_data_kv
        self.activation = nn.GELU()
        if use_alibi:
            self.register_buffer("m", get_alibi_slopes(self.num_heads))

    def forward_packed(self, src, cu_src, max_src, keep_x=True, **_):
        if ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            s = _as_stream(src)
            a = self.mixer.MHA.attend_ln(s, self.norm1, None, cu_src, max_src, None, None)
            x1 = self.mixer.MHA.out_ln(a, s, need_x=False)
            w1, b1, c1 = packed_linear_ln(self.linear_geglu_1, self.norm2, geglu=True)
            hg = ops.gemm_ln_consumer(x1, w1, b1, c1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            return down_projection(hg, w2, b2, s, keep_x)
        src = _as_tensor(src)
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
        x1 = self.mixer.MHA.fused(h, src, cu_src, max_src)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)


class ContextFlashCrossAttentionEncoderLayer(nn.Module):
    """Gene layer without self attention (`only_cross_attention: true`): LN -> cross-MHA(q = x, kv = context,
    un-normalised) -> +src -> LN -> GeGLU -> + src   (reference layers.py:231-325)."""

    def __init__(self, d_model, nhead, hidden_dim=2048, dropout=0.1, batch_first=True, use_alibi=False,
                 make_data_kv=False, mlp_dout=0.0, cross_alibi=False, flash_attn_3=False):
        super().__init__()
        self.crossMHA = FlashAttLayer(d_model, nhead, dropout=dropout, use_alibi=cross_alibi, cross_attn=True,
                                      flash_attn_3=flash_attn_3)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear_geglu_1 = nn.Linear(d_model, hidden_dim)
        self.dropout = nn.Dropout(mlp_dout)
        self.linear_geglu_2 = nn.Linear(hidden_dim // 2, d_model)
        self.use_alibi, self.num_heads, self.make_data_kv = use_alibi, nhead, make_data_kv
        self.activation = nn.GELU()
        if use_alibi:
            self.register_buffer("m", get_alibi_slopes(self.num_heads))

    def forward_packed(self, src, cu_src, max_src, context=None, cu_ctx=None, max_ctx=None, context_kv=None,
                       cu_cross_q=None, max_cross_q=None, keep_x=True):
        if self.make_data_kv:
            # reference layers.py:283-286: Q from the raw context, K / V from norm1(x); context and stream hold the same
            # sequences (the attention output is added to the stream).  Separate-LayerNorm path, as above.
            src, context = _as_tensor(src), _as_tensor(context)
            assert src is not None and context is not None and context.shape == src.shape, \
                "make_data_kv: src and context must have the same shape"
            h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
            kv = self.crossMHA.MHA.project_kv(h)
            x1 = self.crossMHA.MHA.fused(ops.cast16(context), src, cu_src, max_src, kv, cu_src, max_src)
            h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
            w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
            hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)
        cq = cu_src if cu_cross_q is None else cu_cross_q
        mq = max_src if max_cross_q is None else max_cross_q
        if ln_fold_enabled(self.norm1.weight.numel(), self.linear_geglu_2.in_features):
            s = _as_stream(src)
            if context_kv is None:
                context_kv = self.crossMHA.MHA.project_kv_of(context)
            a = self.crossMHA.MHA.attend_ln(s, self.norm1, context_kv, cq, mq, cu_ctx, max_ctx)
            x1 = self.crossMHA.MHA.out_ln(a, s, need_x=False)
            w1, b1, c1 = packed_linear_ln(self.linear_geglu_1, self.norm2, geglu=True)
            hg = ops.gemm_ln_consumer(x1, w1, b1, c1, ops.EPI_GEGLU_BF16)
            w2, b2 = packed_linear(self.linear_geglu_2)
            return down_projection(hg, w2, b2, s, keep_x)
        src, context = _as_tensor(src), _as_tensor(context)
        h = ops.layernorm(src, self.norm1.weight, self.norm1.bias)
        if context_kv is None:
            context_kv = self.crossMHA.MHA.project_kv(ops.cast16(context))
        x1 = self.crossMHA.MHA.fused(h, src, cq, mq, context_kv, cu_ctx, max_ctx)
        h = ops.layernorm(x1, self.norm2.weight, self.norm2.bias)
        w1, b1 = packed_linear(self.linear_geglu_1, geglu=True)
        hg = ops.gemm(h, w1, b1, ops.EPI_GEGLU_BF16)
        w2, b2 = packed_linear(self.linear_geglu_2)
        return ops.gemm(hg, w2, b2, ops.EPI_RES_F32, residual=src)


class StartToken(nn.Module):
    """One learned token in front of every gene sequence (gene_pooling="start_token", reference layers.py:491-499)."""

    def __init__(self, emb_dim):
        super().__init__()
        self.start_token = nn.Parameter(torch.randn(1, 1, emb_dim))

    def forward(self, x):
        return torch.ones(x.size(0), 1, x.size(2), device=x.device) * self.start_token


class AddContext(nn.Module):
    """Tissue embedding added to every CRE token (add_context_to_cres, reference layers.py:558-573)."""

    def __init__(self, num_tissues, emb_dim):
        super().__init__()
        self.num_registry_tokens = num_tissues
        self.registry_tokens = nn.Embedding(num_tissues, emb_dim)

    def get_registry_tokens(self):
        return self.registry_tokens.weight


class MultiRegistry(nn.Module):
    """One registry token per tissue, prepended to the gene tokens (reference layers.py:502-524)."""

    def __init__(self, num_tissues, emb_dim):
        super().__init__()
        self.num_registry_tokens = num_tissues
        self.registry_tokens = nn.Embedding(num_tissues, emb_dim)

    def forward(self, x, tissue_vector):
        # vectorised form of the reference's per-row list comprehension (:509-515)
        reg = self.registry_tokens.weight[tissue_vector[:, 0].long()].unsqueeze(1).to(x.dtype)
        combined = torch.cat((reg, x), dim=1)
        return combined, combined.clone()

    def get_registry_tokens(self):
        return self.registry_tokens.weight


class TissueExpressionHeads(nn.Module):
    """Expression heads (reference layers.py:1012-1144), same module tree / state-dict keys for every combination of
    head_type ("mlp" | "linear"), use_bigger_head and multi_head (one head per tissue in a ModuleDict keyed by the
    tissue id, or one shared head).  The shipped configuration is the shared 'bigger' MLP (:1078-1087):
    Linear -> LayerNorm -> GELU -> Dropout -> Linear -> GELU -> Linear(D,1) -> Softplus.
    Rows are evaluated in batched passes (all rows for a shared head, the rows of one tissue per launch for
    multi_head) instead of the reference's per-row loop with .item() syncs (:1127-1142)."""

    def __init__(self, emb_dim, num_tissues, use_bigger_head=False, multi_head=True, mlp_dout=0.1,
                 loss_fn="poisson", head_type="mlp"):
        super().__init__()
        self.multi_head = multi_head
        self.softplus = loss_fn == "poisson"
        out_act = lambda: nn.Softplus() if self.softplus else nn.Identity()  # noqa: E731
        if head_type == "linear":
            make = lambda: nn.Sequential(nn.Linear(emb_dim, 1), out_act())  # noqa: E731
            self.kind = "linear"
        elif head_type == "mlp":
            if use_bigger_head:
                make = lambda: nn.Sequential(  # noqa: E731
                    nn.Linear(emb_dim, emb_dim), nn.LayerNorm(emb_dim), nn.GELU(), nn.Dropout(mlp_dout),
                    nn.Linear(emb_dim, emb_dim), nn.GELU(), nn.Linear(emb_dim, 1), out_act())
                self.kind = "mlp_big"
            else:
                make = lambda: nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.GELU(), nn.Linear(emb_dim, 1), out_act())  # noqa: E731
                self.kind = "mlp_small"
        else:
            raise ValueError(f"Invalid head type: {head_type}")
        if multi_head:
            self.tissue_expressions = nn.ModuleDict({str(t): make() for t in range(num_tissues)})
        else:
            self.tissue_expressions = make()

    def _apply_head(self, te, g_exp):
        """One head on fp32 rows [n, D] -> fp32 [n, 1]."""
        if self.kind == "linear":
            return ops.rowdot_softplus(g_exp, te[0].weight.reshape(-1).contiguous(), te[0].bias, self.softplus)
        x = ops.cast16(g_exp)
        if self.kind == "mlp_small":
            w0, b0 = packed_linear(te[0])
            h = ops.gemm(x, w0, b0, ops.EPI_GELU_F32)
            return ops.rowdot_softplus(h, te[2].weight.reshape(-1).contiguous(), te[2].bias, self.softplus)
        w0, b0 = packed_linear(te[0])
        h = ops.gemm(x, w0, b0, ops.EPI_F32)
        h = ops.layernorm(h, te[1].weight, te[1].bias, gelu=True)
        w4, b4 = packed_linear(te[4])
        h = ops.gemm(h, w4, b4, ops.EPI_GELU_F32)
        return ops.rowdot_softplus(h, te[6].weight.reshape(-1).contiguous(), te[6].bias, self.softplus)

    def forward(self, g_exp, tissue_vector=None):
        """g_exp fp32 [rows, D] -> fp32 [rows, 1].  tissue_vector: tissue id of every row (needed for multi_head only;
        a tensor [rows] / [rows, 1] or a flat list)."""
        g_exp = g_exp.float().contiguous()
        if not self.multi_head:
            return self._apply_head(self.tissue_expressions, g_exp)
        assert tissue_vector is not None, "multi_head needs the tissue id of every row"
        ids = torch.as_tensor(tissue_vector).reshape(g_exp.shape[0], -1)[:, 0].cpu().tolist()
        out = torch.empty((g_exp.shape[0], 1), dtype=torch.float32, device=g_exp.device)
        for t in sorted(set(ids)):
            rows = torch.tensor([r for r, v in enumerate(ids) if v == t], dtype=torch.int64, device=g_exp.device)
            out[rows] = self._apply_head(self.tissue_expressions[str(int(t))], ops.gather_rows_f32(g_exp, None, rows))
        return out
