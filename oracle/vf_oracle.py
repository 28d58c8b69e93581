"""CPU oracle for the VariantFormer hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (variantformer_amd/) never does and fails loudly without its
HIP library.

What it is: a plain-PyTorch (CPU, fp32) functional restatement of the reference's
algorithm for the hot path, operating directly on a state dict with the
reference's key names.  Each function cites the reference file:line it follows
(paths relative to /root/reference).

How it is pinned:
  * tests/test_oracle_golden.py checks it against tests/golden/*.npz, which were produced
    by running the reference's own classes (fp32, CPU) via tests/golden/make_golden.py;
    agreement is to fp32 round-off (<= 2e-5 relative).
  * The attention arithmetic inside flash_attn.modules.mha.MHA (flash-attn v2.8.3,
    third-party, CUDA-only, absent from /root/reference and from this image) is restated
    from its published semantics: softmax(Q K^T / sqrt(dh) - slope_h * |i - j|) V,
    non-causal, Wqkv rows ordered (three, head, dh), Wkv rows (two, head, dh).
    The fixtures were generated with the same restatement standing in for flash-attn, so
    AT THE ATTENTION BOUNDARY PARITY IS UNPINNED by anything runnable offline.

Two arithmetic modes:
  rounding=None    pure fp32 (what the golden fixtures hold)
  rounding="bf16"  fp32 residual stream / LayerNorm / softmax / accumulation, with operands
                   rounded to bf16 exactly where the HIP kernels round them (DESIGN.md
                   "Rounding points").  This is the checker for the GPU path: kernels and
                   oracle then differ only by accumulation order and exp/erf ulps.
  rounding="fp16"  the same rounding points with IEEE half (the kernels' fp16-operand mode,
                   the reference's 16-mixed / fp16 flash-attn path).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def alibi_slopes(n: int) -> list[float]:
    """seq2gene/modules/layers.py:15-37 (identical to seq2reg/modules.py:13-33 and to
    flash_attn's own helper)."""
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n - c]


def positional_encoding_1d(d_model: int, length: int) -> torch.Tensor:
    """seq2reg/model.py:15-37."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(0, length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


def precision2dtype(precision_str: str) -> torch.dtype:
    """utils/functions.py:12-32."""
    s = precision_str.lower().strip()
    if "bf16" in s:
        return torch.bfloat16
    if "16" in s:
        return torch.float16
    if "32" in s:
        return torch.float32
    raise ValueError(f"Unknown precision string: {precision_str}")


def unpad_input(hidden: torch.Tensor, keep_mask: torch.Tensor):
    """flash_attn.bert_padding.unpad_input [3p] as used at model_combined_modulator.py:179-181:
    returns (packed rows, flat indices int64, cu_seqlens int32 [B+1], max_seqlen, seqlens)."""
    seqlens = keep_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(keep_mask.flatten(), as_tuple=False).flatten()
    cu = F.pad(torch.cumsum(seqlens, 0, dtype=torch.int32), (1, 0))
    flat = hidden.reshape(-1, *hidden.shape[2:])
    return flat[indices], indices, cu, int(seqlens.max()), seqlens


def pad_input(packed: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int):
    """flash_attn.bert_padding.pad_input [3p]: scatter rows into zeros [B,S,...]."""
    out = torch.zeros(batch * seqlen, *packed.shape[1:], dtype=packed.dtype)
    out[indices] = packed
    return out.view(batch, seqlen, *packed.shape[1:])


class LnPending:
    """LayerNorm(x; gamma, beta) not yet applied: the Linear that consumes it evaluates the pair in the folded form of
    the HIP path (see `linear`)."""

    def __init__(self, x, gamma, beta):
        self.x, self.gamma, self.beta = x, gamma, beta

    @property
    def shape(self):
        return self.x.shape


class Rounding:
    """Rounding policy: where the HIP kernels store bf16, the oracle rounds to bf16.  `fold_ln` restates where the bf16
    HIP path rounds around a LayerNorm -> Linear pair: the GEMM reads bf16(x) and bf16(gamma (.) W)
    and the fp32 row statistics are applied to its fp32 accumulator, instead of bf16(LayerNorm(x)) . bf16(W).  Same
    algebra, different rounding points; default = what variantformer_amd does (16-bit operands, VF_LN_FOLD != 0; an fp16
    stream's copy is stored scaled by a power of two, which rounds identically inside the fp16 normal range).
    `res16`: with the fold on, every attention out-projection takes its residual from a 16-bit stream copy -- x1 = attn(..)
    + r(src), x2 = cross(..) + r(x1) -- and those sums' fp32 rows are never stored; a layer's OUTPUT keeps the fp32 layer
    input as its residual (layers.res16_enabled, VF_RES16 != 0).
    `q_prescale`: a cross attention's Wq (and bias) carry softmax_scale * log2(e) before they are rounded, q . k is the
    base-2 logit (layers.q_prescale_enabled, VF_Q_PRESCALE != 0; never without rounding: it is the same function then).
    `trunk16` (needs res16): a layer's output reaches the NEXT LAYER of its stack rounded to the operand type -- where the
    reference's autocast rounds it too (`x = linear_geglu_2(x)` is a 16-bit tensor there and `x += res_long` adds in
    place, layers.py:161-165, seq2reg/modules.py:186-190); the last layer of a stack keeps fp32.  False (default), "s2r"
    (the 6-layer seq2reg encoders only, whose output is mean-pooled over the window and rounded to 16 bits anyway) or
    "all" / True (the 25 + 24 modulator layers too); layers.trunk16_enabled, VF_TRUNK16 = 0 | s2r | 1.
    "f16" (the default): no rounding of the stream itself; each layer's down-projection adds the scaled FP16 copy of its
    input (`trunk` below) -- 11 significant bits, so the trunk's per-layer rounding stays below the operand roundings."""

    def __init__(self, mode: str | None, fold_ln: bool | None = None, res16: bool | None = None,
                 trunk16: bool | None = None, q_prescale: bool | None = None):
        assert mode in (None, "bf16", "fp16")
        self.mode = mode
        import os
        self._trunk16_arg = trunk16
        if q_prescale is None:
            q_prescale = True
        self.q_prescale = bool(q_prescale) and mode is not None
        if fold_ln is None:
            fold_ln = mode is not None and os.environ.get("VF_LN_FOLD", "1") != "0"
        self.fold_ln = bool(fold_ln)
        if res16 is None:
            res16 = True
        self.res16 = bool(res16) and self.fold_ln
        trunk16 = self._trunk16_arg
        if trunk16 is None:
            trunk16 = os.environ.get("VF_TRUNK16", "f16")
        if isinstance(trunk16, str):
            trunk16 = {"0": False, "": False, "s2r": "s2r", "f16": "f16"}.get(trunk16, True)
        if trunk16 is True or trunk16 == 1:
            trunk16 = "all"
        if trunk16 == "f16" and mode == "fp16":
            trunk16 = "all"            # an fp16 operand copy already is the fp16 trunk (layers.trunk_f16_active)
        if trunk16 == "f16" and mode is None:
            trunk16 = False
        self.trunk16 = trunk16 if (trunk16 and self.res16) else False           # False | "s2r" | "all" | "f16"

    def trunk(self, src: torch.Tensor) -> torch.Tensor:
        """The layer input as the residual of the layer's down-projection (x_out = linear_geglu_2(h) + src).  trunk16 ==
        "f16" (the default of variantformer_amd, VF_TRUNK16): every layer of a LayerNorm-folded stack adds the scaled fp16
        copy of its input, fp16(src * 2^-4) * 2^4 (layers.down_projection / vf_gemm_ln_t16) -- the 16-bit OPERAND copy and
        the row statistics still come from the unrounded sum."""
        if self.trunk16 == "f16" and src.shape[-1] % 64 == 0:
            return (src * 0.0625).to(torch.float16).to(torch.float32) * 16.0
        return src

    def res(self, x1: torch.Tensor) -> torch.Tensor:
        """x1 as the residual of the cross-attention out-projection (see `res16`)."""
        return self.r(x1) if (self.res16 and x1.shape[-1] % 64 == 0) else x1

    def out(self, x: torch.Tensor, last: bool = False, s2r: bool = False) -> torch.Tensor:
        """A layer's output (the trunk) on its way to the next layer of the stack: fp32, or rounded to the operand type
        (`trunk16`: "s2r" = in the seq2reg encoders only, "all" = in the modulator stacks too).  The `last` layer of a stack
        keeps its fp32 result (it is pooled / returned, not fed to a layer)."""
        on = self.trunk16 == "all" or (self.trunk16 == "s2r" and s2r)
        return self.r(x) if (on and not last and x.shape[-1] % 64 == 0) else x

    def r(self, x: torch.Tensor) -> torch.Tensor:
        if self.mode is None:
            return x
        return x.to(torch.bfloat16 if self.mode == "bf16" else torch.float16).to(torch.float32)

    def ln(self, x, w, b, fold: bool = True):
        """LayerNorm whose only consumers are Linear layers: rounded output, or the pending (folded) form."""
        if self.fold_ln and fold and x.shape[-1] % 64 == 0:
            return LnPending(x, w, b)
        return self.r(layer_norm(x, w, b))


def linear(x, w, b, rnd: Rounding, wscale: float = 1.0):
    """nn.Linear under the kernel contract: bf16 operands (x already rounded by its
    producer, w rounded once at load), fp32 accumulate, fp32 bias.
    With a pending LayerNorm:  LN(x) W^T + b = rstd * (x W'^T - mean * rowsum(W')) + (W beta + b),  W' = gamma (.) W,
    operands x and W' rounded, statistics and the correction in fp32.
    `wscale`: the projection multiplied by a constant (a float, or one factor per output row [N, 1]) that is folded into
    the weights BEFORE they are rounded (and into the fp32 bias): Rounding.q_prescale."""
    scaled = None
    if torch.is_tensor(wscale):
        scaled = wscale.to(torch.float32).reshape(-1, 1)
    elif wscale != 1.0:
        scaled = torch.full((w.shape[0], 1), float(wscale))
    if isinstance(x, LnPending):
        xs = x.x
        mean = xs.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(xs.var(dim=-1, unbiased=False, keepdim=True) + 1e-5)
        wg = w * x.gamma[None, :]
        wp = rnd.r(wg if scaled is None else wg * scaled)
        bias = w @ x.beta
        if b is not None:
            bias = bias + b
        if scaled is not None:
            bias = bias * scaled[:, 0]
        return (F.linear(rnd.r(xs), wp) - mean * wp.sum(dim=1)[None, :]) * rstd + bias
    if scaled is not None:
        return F.linear(rnd.r(x), rnd.r(w * scaled), None if b is None else b * scaled[:, 0])
    return F.linear(rnd.r(x), rnd.r(w), b)


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def attention(q, k, v, slopes, rnd: Rounding, q_log2: bool = False):
    """Textbook restatement of flash-attn's varlen forward for ONE sequence [3p].
    q [sq,H,dh], k/v [sk,H,dh] (already rounded to the storage dtype by their producer).
    scores fp32, ALiBi bias -slope*|i + (sk - sq) - j|, softmax fp32; P is rounded to the
    operand dtype before PV while the normaliser is summed from the unrounded P.
    q_log2: q already carries softmax_scale * log2(e) (Rounding.q_prescale): q . k is the base-2 logit."""
    dh = q.shape[-1]
    if q_log2:
        # base-2 logits as the matrix pipe delivers them; the kernels exponentiate them directly (or, for rows outside the
        # representable range, after an INTEGER offset: a power of two does not move a rounding), round P to the operand
        # type and divide by the sum of the ROUNDED P (the denominator runs on the matrix pipe next to P . V)
        s2 = torch.einsum("qhd,khd->hqk", q, k)
        if slopes is not None:
            sq, sk = q.shape[0], k.shape[0]
            i = torch.arange(sq)[:, None] + (sk - sq)
            j = torch.arange(sk)[None, :]
            s2 = s2 - (slopes.to(s2.dtype) * math.log2(math.e))[:, None, None] * (i - j).abs().to(s2.dtype)[None]
        pr = rnd.r(torch.exp2(s2 - torch.ceil(s2.max(dim=-1, keepdim=True).values)))
        o = torch.einsum("hqk,khd->hqd", pr, v) / pr.sum(dim=-1, keepdim=True)
        return o.permute(1, 0, 2)
    s = torch.einsum("qhd,khd->hqk", q, k) * (1.0 / math.sqrt(dh))
    if slopes is not None:
        sq, sk = q.shape[0], k.shape[0]
        i = torch.arange(sq)[:, None] + (sk - sq)
        j = torch.arange(sk)[None, :]
        s = s - slopes.to(s.dtype)[:, None, None] * (i - j).abs().to(s.dtype)[None]
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    o = torch.einsum("hqk,khd->hqd", rnd.r(p), v) / l
    return o.permute(1, 0, 2)          # [sq,H,dh]


def attention_counted(q, k_tab, v_tab, counts, q_log2: bool):
    """Attention of q [sq, H, dh] against keys that are copies of the C distinct rows k_tab / v_tab [C, H, dh], row c occurring
    counts[c] times: softmax over the repeated keys = softmax over the distinct rows with log(count) added to the logit.  The
    same function as attention() on the expanded rows; restates where vf_attn_counted_keys rounds: fp32 scores and weights,
    no 16-bit rounding of P (the caller rounds the output once)."""
    dh = q.shape[-1]
    s2 = torch.einsum("qhd,chd->hqc", q, k_tab) * (1.0 if q_log2 else math.log2(math.e) / math.sqrt(dh))
    s2 = s2 + torch.log2(counts.to(s2.dtype))[None, None, :]             # -inf for a label the sequence does not hold
    p = torch.exp2(s2 - s2.max(dim=-1, keepdim=True).values)
    o = torch.einsum("hqc,chd->hqd", p, v_tab) / p.sum(dim=-1, keepdim=True)
    return o.permute(1, 0, 2)


def geglu_ffn(x, sd, pfx, rnd: Rounding):
    """LN'd input -> Linear(d,2048) -> a * gelu(gate) -> Linear(1024,d)
    (layers.py:159-162, seq2reg/modules.py:184-187; erf GELU)."""
    h = linear(x, sd[pfx + "linear_geglu_1.weight"], sd[pfx + "linear_geglu_1.bias"], rnd)
    a, g = h.chunk(2, dim=-1)
    h = rnd.r(a * F.gelu(g))
    return linear(h, sd[pfx + "linear_geglu_2.weight"], sd[pfx + "linear_geglu_2.bias"], rnd)


def mha_self(x, sd, pfx, H, cu, slopes, rnd: Rounding):
    """flash_attn MHA self path [3p] on a packed stream x [tokens, D] with cu_seqlens."""
    D = x.shape[-1]
    dh = D // H
    pre = rnd.q_prescale                 # the Q rows of Wqkv carry the base-2 softmax scale before their rounding
    ws = 1.0
    if pre:
        ws = torch.ones(3 * D, 1)
        ws[:D] = math.log2(math.e) / math.sqrt(dh)
    qkv = rnd.r(linear(x, sd[pfx + "Wqkv.weight"], sd[pfx + "Wqkv.bias"], rnd, wscale=ws)).view(-1, 3, H, dh)
    out = torch.empty(x.shape[0], D)
    for b in range(len(cu) - 1):
        a, e = int(cu[b]), int(cu[b + 1])
        if e > a:
            out[a:e] = attention(qkv[a:e, 0], qkv[a:e, 1], qkv[a:e, 2], slopes, rnd, q_log2=pre).reshape(e - a, D)
    out = rnd.r(out)
    return linear(out, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"], rnd)


def mha_cross(xq, xkv, sd, pfx, H, cu_q, cu_k, rnd: Rounding, slopes=None, kv_labels=None):
    """flash_attn MHA cross path [3p]: Wq on the query stream, Wkv on the context stream;
    no ALiBi in the shipped configuration (cross_alibi: false, configs/vf_model.yaml:13).
    kv_labels int64 [tokens_k] (the CRE layers' context cross attention, whose context row of a key is a function of its
    label alone): under a rounding mode the keys are taken as the distinct rows with their counts (attention_counted: the
    form vf_attn_counted_keys evaluates); pure fp32 arithmetic keeps the reference's expanded form -- the same function."""
    D = xq.shape[-1]
    dh = D // H
    pre = rnd.q_prescale                 # the softmax scale, in base 2, folded into Wq before its rounding (layers.q_prescale_enabled)
    q = rnd.r(linear(xq, sd[pfx + "Wq.weight"], sd[pfx + "Wq.bias"], rnd,
                     wscale=math.log2(math.e) / math.sqrt(dh) if pre else 1.0)).view(-1, H, dh)
    kv = rnd.r(linear(xkv, sd[pfx + "Wkv.weight"], sd[pfx + "Wkv.bias"], rnd)).view(-1, 2, H, dh)
    out = torch.empty(xq.shape[0], D)
    counted = kv_labels is not None and rnd.mode is not None and slopes is None and COUNTED_CONTEXT_KEYS
    if counted and LOWRANK_CONTEXT and any((H * c) % 64 == 0 for c in range(9 + 1, 17, 2)):
        # LOW-RANK form (layers.MHA.lowrank_tables / cross_lowrank): logits = LN(x) . (scale Wq_h^T k_c) as ONE rounded-operand
        # GEMM, 16-bit softmax weights over the distinct labels (log2 count added), out_proj folded into w . (Wo_h v_c); k, v of
        # the distinct labels in fp32 (no 16-bit q / k / v).  Restates where the two skinny GEMMs round.
        labs = sorted(set(kv_labels.tolist()))
        first = [int((kv_labels == l).nonzero()[0]) for l in labs]
        kvf = (xkv[first] @ sd[pfx + "Wkv.weight"].t() + sd[pfx + "Wkv.bias"]).view(len(labs), 2, H, dh)
        sc_ = math.log2(math.e) / math.sqrt(dh)
        z = torch.einsum("chd,hdk->hck", kvf[:, 0], sd[pfx + "Wq.weight"].view(H, dh, D)) * sc_            # [H, C, D]
        zb = torch.einsum("chd,hd->hc", kvf[:, 0], sd[pfx + "Wq.bias"].view(H, dh)) * sc_
        u = torch.einsum("nhd,chd->nhc", sd[pfx + "out_proj.weight"].view(D, H, dh), kvf[:, 1])           # [D, H, C]
        C = len(labs)
        s2 = linear(xq, z.reshape(H * C, D), zb.reshape(H * C), rnd).view(-1, H, C)
        w = torch.zeros_like(s2)
        for b in range(len(cu_q) - 1):
            a, e = int(cu_q[b]), int(cu_q[b + 1])
            ka, ke = int(cu_k[b]), int(cu_k[b + 1])
            if e > a:
                cnt = torch.tensor([float((kv_labels[ka:ke] == l).sum()) for l in labs])
                t = s2[a:e] + torch.log2(cnt)[None, None, :]
                p = torch.exp2(t - t.max(dim=-1, keepdim=True).values)
                w[a:e] = rnd.r(p / p.sum(dim=-1, keepdim=True))
        return linear(w.reshape(-1, H * C), u.reshape(D, H * C), sd[pfx + "out_proj.bias"], rnd)
    for b in range(len(cu_q) - 1):
        a, e = int(cu_q[b]), int(cu_q[b + 1])
        ka, ke = int(cu_k[b]), int(cu_k[b + 1])
        if e > a and counted and ke > ka:
            lab = kv_labels[ka:ke]
            uniq, first = [], []
            for i, l in enumerate(lab.tolist()):                       # distinct labels, with the row of their first occurrence
                if l not in uniq:
                    uniq.append(l)
                    first.append(ka + i)
            counts = torch.tensor([int((lab == l).sum()) for l in uniq], dtype=torch.float32)
            out[a:e] = attention_counted(q[a:e], kv[first, 0], kv[first, 1], counts, pre).reshape(e - a, D)
        elif e > a:
            out[a:e] = attention(q[a:e], kv[ka:ke, 0], kv[ka:ke, 1], slopes, rnd, q_log2=pre).reshape(e - a, D)
    out = rnd.r(out)
    return linear(out, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"], rnd)


# --------------------------------------------------------------------------------------
# seq2reg (a-4, a-5)
# --------------------------------------------------------------------------------------
@dataclass
class Seq2RegHP:
    embedding_dim: int
    num_heads: int
    num_layers: int
    token_length: int
    positional_encoding: str = "sinusoidal"
    seq_pool: str = "mean"
    use_context: bool = False
    expand_context: bool = False

    @classmethod
    def from_hparams(cls, hp: dict):
        return cls(hp["embedding_dim"], hp["num_heads"], hp["num_layers"], hp["token_length"],
                   hp.get("positional_encoding", "sinusoidal"), hp.get("seq_pool", "mean"),
                   hp.get("use_context", False), hp.get("expand_context", False))


def seq2reg_layer(x, cu, sd, pfx, hp: Seq2RegHP, slopes, rnd: Rounding, last=False, pool_cu=None):
    """FlashTransformerLayer.forward (seq2reg/modules.py:149-191) on the packed valid tokens.
    Pad positions never influence valid ones (attention runs on the unpadded stream, :159-171;
    everything else is per-token) and are excluded from the pool, so only valid tokens are kept."""
    h = rnd.ln(x, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    a = mha_self(h, sd, pfx + "MHA.", hp.num_heads, cu, slopes, rnd)
    x1 = a + rnd.res(x)                                              # :179  x += res_short (16-bit copy: Rounding.res16)
    h = rnd.ln(x1, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    if pool_cu is not None:
        # the mean pool of seq2reg/model.py:263-267 taken BEFORE linear_geglu_2 (which is linear, seq2reg/modules.py:186-188):
        # mean(W2 h + b + src) = W2 mean(h) + b + mean(src) -- what variantformer_amd executes for the encoder's last layer
        # (FlashTransformerLayer._pooled_down_projection); mean(h) is NOT rounded again (it passes the GEMM as hi + lo)
        hh = linear(h, sd[pfx + "linear_geglu_1.weight"], sd[pfx + "linear_geglu_1.bias"], rnd)
        a_, g_ = hh.chunk(2, dim=-1)
        hg, res = rnd.r(a_ * F.gelu(g_)), rnd.trunk(x)
        W = len(pool_cu) - 1
        out = torch.full((W, x.shape[1]), float("nan"))
        w2, b2 = rnd.r(sd[pfx + "linear_geglu_2.weight"]), sd[pfx + "linear_geglu_2.bias"]
        for w in range(W):
            a, e = int(pool_cu[w]), int(pool_cu[w + 1])
            if e > a:
                out[w] = F.linear(hg[a:e].mean(dim=0), w2, b2) + res[a:e].mean(dim=0)
        return out
    return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(x), last, s2r=True)                  # :188  x += res_long (= layer input)


def seq2reg_context_layer(x, ctx, cu, sd, pfx, hp: Seq2RegHP, slopes, rnd: Rounding, last=False):
    """seq2reg's ContextFlashAttentionEncoderLayer.forward (seq2reg/modules.py:72-126), make_data_kv false: LN1 ->
    self-MHA -> +src -> LN2 -> cross-MHA(q = x, kv = context rows of the same window, no ALiBi, same key padding as the
    tokens :105-112) -> +res_short -> LN3 -> GeGLU -> + src."""
    h = rnd.ln(x, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    x1 = mha_self(h, sd, pfx + "mixer.MHA.", hp.num_heads, cu, slopes, rnd) + rnd.res(x)
    h = rnd.ln(x1, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    x2 = mha_cross(h, ctx, sd, pfx + "crossMHA.MHA.", hp.num_heads, cu, cu, rnd) + rnd.res(x1)
    h = rnd.ln(x2, sd[pfx + "norm3.weight"], sd[pfx + "norm3.bias"])
    return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(x), last, s2r=True)


def seq2reg_embed(ids, pad_mask, sd, pfx, hp: Seq2RegHP, rnd: Rounding, context=None, pool_before_down=False):
    """Seq2RegPredictor.forward(only_embed=True) (seq2reg/model.py:193-279).
    ids int64 [b,1,L], pad_mask bool [b,1,L] (True = pad), context int64 [b] (use_context) -> [b,1,d].
    pool_before_down (mean pool, no context layers): the exact re-ordering variantformer_amd executes -- the pool in front of
    the last layer's linear_geglu_2 (seq2reg_layer(pool_cu=...)); the default is the reference's order of operations."""
    b, ns, L = ids.shape
    ids = ids.reshape(b * ns, L)
    pad = pad_mask.reshape(b * ns, L)
    x = sd[pfx + "token_embedding.weight"][ids]                       # :215
    if hp.positional_encoding == "sinusoidal":
        x = x + positional_encoding_1d(hp.embedding_dim, hp.token_length)   # :219-220
        slopes = None
    else:
        slopes = torch.tensor(alibi_slopes(hp.num_heads), dtype=torch.float32)
    xp, idx, cu, _, lens = unpad_input(x, ~pad)
    d = hp.embedding_dim
    if hp.use_context:                                                # :222-245
        c = sd[pfx + "context_embedding.weight"][context.long().reshape(-1)].repeat_interleave(ns, dim=0)   # [b*ns, d]
        if hp.expand_context:       # Linear(1, token_length) on every embedding element: context[w, p, :] = c * W[p] + B[p]
            We, Be = sd[pfx + "expand_context.weight"][:, 0], sd[pfx + "expand_context.bias"]
            cfull = c[:, None, :] * We[None, :, None] + Be[None, :, None]
        else:
            cfull = c[:, None, :].expand(b * ns, L, d)
        ctx = cfull.reshape(b * ns * L, d)[idx]                       # context rows of the valid tokens
        for l in range(hp.num_layers):
            xp = seq2reg_context_layer(xp, ctx, cu, sd, f"{pfx}transformer_encoder.{l}.", hp, slopes, rnd,
                                       last=l + 1 == hp.num_layers)
    else:
        early = pool_before_down and hp.seq_pool == "mean"
        for l in range(hp.num_layers):
            lastl = l + 1 == hp.num_layers
            xp = seq2reg_layer(xp, cu, sd, f"{pfx}transformer_encoder.{l}.", hp, slopes, rnd, last=lastl,
                               pool_cu=cu if (early and lastl) else None)
        if early:
            return xp.view(b, ns, d)
    if hp.seq_pool == "mean":                                         # :263-267
        out = torch.zeros(b * ns, d)
        for w in range(b * ns):
            a, e = int(cu[w]), int(cu[w + 1])
            out[w] = xp[a:e].sum(dim=0) / float(e - a) if e > a else float("nan")
    elif hp.seq_pool == "max":                                        # :257-261
        out = torch.full((b * ns, d), float("-inf"))
        for w in range(b * ns):
            a, e = int(cu[w]), int(cu[w + 1])
            if e > a:
                out[w] = xp[a:e].max(dim=0).values
    else:                                                             # "linear" :268-272: Linear(token_length, 1) over the
        full = pad_input(xp, idx, b * ns, L)                          # token axis of the zero-masked window
        out = torch.einsum("wpd,p->wd", full, sd[pfx + "linear.weight"][0]) + sd[pfx + "linear.bias"][0]
    return out.view(b, ns, d)


# --------------------------------------------------------------------------------------
# seq2gene (a-2, a-3, a-6 .. a-13)
# --------------------------------------------------------------------------------------
@dataclass
class Seq2GeneHP:
    emb_dim: int
    num_heads: int
    num_layers: int
    token_dim: int
    gene_emb_dim: int
    num_tissues: int = 63
    use_alibi: bool = True
    extras: dict = field(default_factory=dict)
    # options the shipped configuration leaves at these values (configs/vf_model.yaml:12-37)
    only_cross_attention: bool = False      # gene layers without self attention (layers.py:231-325)
    use_res: bool = False                   # gene-stream input added back after every gene layer (:236,253,284)
    cross_alibi: bool = False               # ALiBi on the cross attentions too (layers.py:60-71)
    add_context_to_cres: bool = False       # tissue embedding added to the CRE tokens (:669-670, layers.py:558-573)
    gene_pooling: str = "multi_registry"    # or "start_token" / "max" (:330-396; "mean" does not reduce in the reference)
    use_context: bool = True                # False: context-free CRE layers (FlashAttentionEncoderLayer, layers.py:168-228)
    multi_head: bool = False                # one expression head per tissue (layers.py:1042-1050,1060-1076,1090-1102)
    use_bigger_head: bool = True
    head_type: str = "mlp"                  # or "linear" (layers.py:1040-1055)

    @property
    def shipped(self) -> bool:
        return not (self.only_cross_attention or self.use_res or self.cross_alibi or self.add_context_to_cres) \
            and self.gene_pooling == "multi_registry"

    @classmethod
    def from_kwargs(cls, kw: dict):
        pooling = kw.get("gene_pooling")
        assert pooling in ("multi_registry", "start_token", "max"), pooling
        return cls(kw["emb_dim"], kw["num_heads"], kw["num_layers"], kw["token_dim"], kw["gene_emb_dim"],
                   kw.get("num_tissues", 63), kw.get("use_alibi", True),
                   only_cross_attention=kw.get("only_cross_attention", True), use_res=kw.get("use_res", False),
                   cross_alibi=kw.get("cross_alibi", False) and kw.get("use_alibi", True),
                   add_context_to_cres=kw.get("add_context_to_cres", False), gene_pooling=pooling,
                   use_context=kw.get("use_context", False), multi_head=kw.get("multi_head", True),
                   use_bigger_head=kw.get("use_bigger_head", False), head_type=kw.get("head_type", "mlp"))


COUNTED_CONTEXT_KEYS = True       # tests flip it together with layers.COUNTED_CONTEXT_KEYS
LOWRANK_CONTEXT = True            # ... and layers.LOWRANK_CONTEXT


def modulator_layer(src, ctx, cu_src, cu_ctx, sd, pfx, H, slopes, rnd: Rounding, cross_slopes=None, last=False,
                    make_data_kv=False, ctx_labels=None):
    """ContextFlashAttentionEncoderLayer.forward (seq2gene/modules/layers.py:88-165) on packed
    streams: LN1 -> self-MHA(ALiBi) -> +src -> LN2 -> cross-MHA(q = x, kv = ctx RAW, no norm)
    -> +res_short -> LN3 -> GeGLU -> + src (the LAYER INPUT, :99,163).
    make_data_kv (:133-136; no reference call site enables it): q = ctx RAW, kv = LN2(x); ctx holds the stream's sequences."""
    h = rnd.ln(src, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    x1 = mha_self(h, sd, pfx + "mixer.MHA.", H, cu_src, slopes, rnd) + rnd.res(src)
    h = rnd.ln(x1, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    if make_data_kv:
        x2 = mha_cross(ctx, h, sd, pfx + "crossMHA.MHA.", H, cu_ctx, cu_src, rnd, cross_slopes) + rnd.res(x1)
        h = rnd.ln(x2, sd[pfx + "norm3.weight"], sd[pfx + "norm3.bias"])
        return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(src), last)
    x2 = mha_cross(h, ctx, sd, pfx + "crossMHA.MHA.", H, cu_src, cu_ctx, rnd, cross_slopes, kv_labels=ctx_labels) + rnd.res(x1)
    h = rnd.ln(x2, sd[pfx + "norm3.weight"], sd[pfx + "norm3.bias"])
    return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(src), last)


def self_only_layer(src, cu_src, sd, pfx, H, slopes, rnd: Rounding, last=False):
    """FlashAttentionEncoderLayer.forward (layers.py:168-228): LN1 -> self-MHA(ALiBi) -> +src -> LN2 -> GeGLU -> + src
    (norm3 is constructed but never applied)."""
    h = rnd.ln(src, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    x1 = mha_self(h, sd, pfx + "mixer.MHA.", H, cu_src, slopes, rnd) + rnd.res(src)
    h = rnd.ln(x1, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(src), last)


def cre_layer(cre, ctx, cu_cre, sd, pfx, hp, slopes, rnd: Rounding, last=False, ctx_labels=None):
    """One CRE layer: with the second-level context (:262-270) or context-free (:271-274).  Never cross_alibi (:78-88).
    ctx_labels: the label whose embedding each context row is (ctx = Embedding(9)[ctx_labels], :168)."""
    if hp.use_context:
        return modulator_layer(cre, ctx, cu_cre, cu_cre, sd, pfx, hp.num_heads, slopes, rnd, last=last, ctx_labels=ctx_labels)
    return self_only_layer(cre, cu_cre, sd, pfx, hp.num_heads, slopes, rnd, last=last)


def cross_only_layer(src, ctx, cu_src, cu_ctx, sd, pfx, H, rnd: Rounding, cross_slopes=None, last=False,
                     make_data_kv=False):
    """ContextFlashCrossAttentionEncoderLayer.forward (layers.py:231-325): LN1 -> cross-MHA(q = x, kv = ctx raw)
    -> +src -> LN2 -> GeGLU -> + src (the layer input).  make_data_kv (:283-286): q = ctx raw, kv = LN1(x)."""
    h = rnd.ln(src, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    if make_data_kv:
        x1 = mha_cross(ctx, h, sd, pfx + "crossMHA.MHA.", H, cu_ctx, cu_src, rnd, cross_slopes) + rnd.res(src)
    else:
        x1 = mha_cross(h, ctx, sd, pfx + "crossMHA.MHA.", H, cu_src, cu_ctx, rnd, cross_slopes) + rnd.res(src)
    h = rnd.ln(x1, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    return rnd.out(geglu_ffn(h, sd, pfx, rnd) + rnd.trunk(src), last)


def combined_modulator(cre_x, gene_x, ctx_labels, cu_cre, cu_gene, sd, pfx, hp: Seq2GeneHP, rnd: Rounding,
                       collect=None):
    """CombinedModulator.forward (model_combined_modulator.py:137-328) on packed streams.
    cre_x [sum N, D], gene_x [sum G, D], ctx_labels int64 [sum N]."""
    slopes = torch.tensor(alibi_slopes(hp.num_heads), dtype=torch.float32) if hp.use_alibi else None
    xs = slopes if hp.cross_alibi else None
    ctx = sd[pfx + "second_level_context_embedding.weight"][ctx_labels] if hp.use_context else None   # :166-168

    nl = hp.num_layers

    def gene_layer(g, c, p, li):
        last = li == nl - 1 or hp.use_res        # a stream `use_res` adds to keeps its fp32 rows (Rounding.out)
        if hp.only_cross_attention:                                               # :106-119
            return cross_only_layer(g, c, cu_gene, cu_cre, sd, p, hp.num_heads, rnd, xs, last=last)
        return modulator_layer(g, c, cu_gene, cu_cre, sd, p, hp.num_heads, slopes, rnd, xs, last=last)

    cre, gene = cre_x, gene_x
    gene_res = gene_x if hp.use_res else None                                    # :236
    gene = gene_layer(gene, cre, pfx + "gene_layers.0.", 0)                      # :244-250
    if collect is not None:
        collect["first_gene_layer_out"] = gene.clone()                          # the layer's own output
    if gene_res is not None:
        gene = gene + gene_res                                                   # :253-254
    for i in range(hp.num_layers - 1):                                           # :258-285
        # the CRE layers are built without cross_alibi (:78-88): their context cross attention never has a bias
        cre = cre_layer(cre, ctx, cu_cre, sd, f"{pfx}cre_layers.{i}.", hp, slopes, rnd, last=i == nl - 2,
                        ctx_labels=ctx_labels if hp.use_context else None)
        if collect is not None and i == 0:
            collect["first_cre_layer_out"] = cre.clone()
        gene = gene_layer(gene, cre, f"{pfx}gene_layers.{i + 1}.", i + 1)
        if gene_res is not None:
            gene = gene + gene_res
    return gene, cre


def _one_head(emb, sd, p, hp, rnd: Rounding):
    """One expression head (layers.py:1040-1110).  Kernel contract: the D x D Linears are bf16-operand GEMMs; the final
    D -> 1 dot product and everything elementwise stay fp32."""
    if hp.head_type == "linear":                                             # Linear(D,1) -> Softplus
        return F.softplus(F.linear(emb, sd[p + "0.weight"], sd[p + "0.bias"]))
    if not hp.use_bigger_head:                                               # Linear -> GELU -> Linear(D,1) -> Softplus
        h = F.gelu(linear(emb, sd[p + "0.weight"], sd[p + "0.bias"], rnd))
        return F.softplus(F.linear(h, sd[p + "2.weight"], sd[p + "2.bias"]))
    h = linear(emb, sd[p + "0.weight"], sd[p + "0.bias"], rnd)              # the shipped 'bigger' head (:1078-1087)
    h = rnd.r(F.gelu(layer_norm(h, sd[p + "1.weight"], sd[p + "1.bias"])))
    h = F.gelu(linear(h, sd[p + "4.weight"], sd[p + "4.bias"], rnd))
    h = F.linear(h, sd[p + "6.weight"], sd[p + "6.bias"])
    return F.softplus(h)


def tissue_head(emb, sd, pfx, rnd: Rounding, hp=None, tissues=None):
    """TissueExpressionHeads.forward (layers.py:1113-1144): the shared head, or with multi_head the head of each row's
    tissue (ModuleDict keyed by the tissue id)."""
    hp = hp if hp is not None else Seq2GeneHP(0, 0, 0, 0, 0)
    p = pfx + "tissue_expressions."
    if not hp.multi_head:
        return _one_head(emb, sd, p, hp, rnd)
    out = torch.empty(emb.shape[0], 1)
    for r, t in enumerate(tissues):
        out[r] = _one_head(emb[r:r + 1], sd, f"{p}{int(t)}.", hp, rnd)[0]
    return out


def forward(batch: dict, sd: dict, cre_hp: Seq2RegHP, gene_hp: Seq2RegHP, hp: Seq2GeneHP,
            rounding: str | None = None, share_cre_stream: bool = False, collect: dict | None = None):
    """Seq2GenePredictorCombinedModulator.forward (model_combined_modulator.py:540-720) for a
    collate_fn_batching dict.  Returns (pred [sum T,1], emb [sum T,D]).

    share_cre_stream=False follows the reference literally: every gene's CRE stream and gene
    stream are repeated once per requested tissue (:622-649).  share_cre_stream=True evaluates
    the tissue-independent CRE stream once per gene -- the exact de-duplication the HIP path
    uses; tests assert both give the same numbers.
    """
    rnd = rounding if isinstance(rounding, Rounding) else Rounding(rounding)
    n_genes = len(batch["cre_sequences"])
    D = hp.emb_dim
    # --- seq2reg over CRE windows and gene chunks (transform_with_batching, :722-829).  The
    # reference chunks by <=1024 windows (:760-785); windows are independent, so chunking does
    # not change any value.
    cre_tok = [seq2reg_embed(batch["cre_sequences"][i], batch["cre_attention_masks"][i], sd,
                             "cre_tokenizer.", cre_hp, rnd)[:, 0, :] for i in range(n_genes)]
    gene_tok = [seq2reg_embed(batch["gene_embeddings"][i], batch["gene_attention_masks"][i], sd,
                              "gene_tokenizer.", gene_hp, rnd)[:, 0, :] for i in range(n_genes)]
    if collect is not None:
        collect["cre_tok"], collect["gene_tok"] = cre_tok, gene_tok
    # --- maps (:610-612)
    if "cre_map.weight" in sd:
        cre_x = [linear(t, sd["cre_map.weight"], sd["cre_map.bias"], rnd) for t in cre_tok]
    else:
        cre_x = cre_tok
    gene_x = [linear(t, sd["gene_map.weight"], sd["gene_map.bias"], rnd) for t in gene_tok]
    assert hp.shipped or not share_cre_stream, "the de-duplicated evaluation exists for the shipped configuration only"

    embs = []
    first_gene, first_cre, mod_out, mod_cre_out = [], [], [], []
    for i in range(n_genes):
        tissues = [int(t) for t in batch["tissue_context"][i]]
        T, N, C = len(tissues), cre_x[i].shape[0], gene_x[i].shape[0]
        labels = batch["ref_cre_labels"][i].long()
        # prepare_input (:330-368): registry token per tissue (MultiRegistry.forward layers.py:508-521), one shared
        # start token (StartToken, layers.py:491-499), or nothing in front of the gene tokens (max pooling)
        if hp.gene_pooling == "multi_registry":
            reg = sd["start_tkn.registry_tokens.weight"]
            g = torch.cat([torch.cat([reg[t][None, :], gene_x[i]], dim=0) for t in tissues], dim=0)   # [T*(C+1), D]
            G = C + 1
        elif hp.gene_pooling == "start_token":
            g = torch.cat([torch.cat([sd["start_tkn.start_token"].reshape(1, D), gene_x[i]], dim=0) for _ in tissues], dim=0)
            G = C + 1
        else:
            g = gene_x[i].repeat(T, 1)
            G = C
        cu_g = torch.arange(0, T + 1, dtype=torch.int32) * G
        col = {} if collect is not None else None
        if share_cre_stream:
            # one CRE stream; all T tissue copies of the gene stream attend to it
            out, cre_out = _modulator_shared(cre_x[i], g, labels, T, C + 1, sd, hp, rnd, col)
            cre_out = cre_out[None].expand(T, N, D)
        else:
            cre_rep = cre_x[i].repeat(T, 1)
            if hp.add_context_to_cres:                       # AddContext (layers.py:558-573): + tissue embedding
                add = sd["add_context.registry_tokens.weight"]
                cre_rep = cre_rep + torch.cat([add[t][None, :].expand(N, D) for t in tissues], dim=0)
            lab_rep = labels.repeat(T)
            cu_c = torch.arange(0, T + 1, dtype=torch.int32) * N
            out, cre_out = combined_modulator(cre_rep, g, lab_rep, cu_c, cu_g, sd, "combined_modulator.", hp, rnd, col)
            cre_out = cre_out.view(T, N, D)
        if collect is not None:
            first_gene.append(col["first_gene_layer_out"])
            first_cre.append(col["first_cre_layer_out"])
            mod_out.append(out.view(T, G, D))
            mod_cre_out.append(cre_out)
        if hp.gene_pooling == "max":
            embs.append(out.view(T, G, D).max(dim=1).values)   # pool_outputs max (:380-389); every chunk is valid here
        else:
            embs.append(out.view(T, G, D)[:, 0, :])            # start / registry token (:391-392)
    if collect is not None:
        collect["first_gene_layer_out"] = torch.cat(first_gene)
        collect["first_cre_layer_out"] = torch.cat(first_cre)
        collect["modulator_gene_out"] = mod_out          # per gene [T, G, D] (G includes the start / registry token)
        collect["modulator_cre_out"] = mod_cre_out        # per gene [T, N, D]: the CRE stream after the last CRE layer
    emb = torch.cat(embs, dim=0)
    all_tissues = [int(t) for i in range(n_genes) for t in batch["tissue_context"][i]]
    pred = tissue_head(emb, sd, "tissue_heads.", rnd, hp, all_tissues)
    return pred, emb


def _modulator_shared(cre_x, gene_x, labels, T, G, sd, hp: Seq2GeneHP, rnd: Rounding, collect=None):
    """De-duplicated evaluation of CombinedModulator.forward for ONE gene: the CRE stream
    (tissue-independent while add_context_to_cres is false, configs/vf_model.yaml:16) runs once;
    gene-layer cross-attention has no positional bias, so the T*G query rows form one query
    block against the gene's N keys."""
    pfx = "combined_modulator."
    H = hp.num_heads
    slopes = torch.tensor(alibi_slopes(H), dtype=torch.float32) if hp.use_alibi else None
    N = cre_x.shape[0]
    ctx = sd[pfx + "second_level_context_embedding.weight"][labels] if hp.use_context else None
    cu_c = torch.tensor([0, N], dtype=torch.int32)
    cu_g = torch.arange(0, T + 1, dtype=torch.int32) * G

    def gene_layer(src, kvsrc, p, last):
        h = rnd.ln(src, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        x1 = mha_self(h, sd, p + "mixer.MHA.", H, cu_g, slopes, rnd) + rnd.res(src)
        h = rnd.ln(x1, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        x2 = mha_cross(h, kvsrc, sd, p + "crossMHA.MHA.", H, torch.tensor([0, T * G], dtype=torch.int32), cu_c, rnd) \
            + rnd.res(x1)
        h = rnd.ln(x2, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
        return rnd.out(geglu_ffn(h, sd, p, rnd) + rnd.trunk(src), last)

    nl = hp.num_layers
    cre, gene = cre_x, gene_x
    gene = gene_layer(gene, cre, pfx + "gene_layers.0.", nl == 1)
    if collect is not None:
        collect["first_gene_layer_out"] = gene.clone()
    for i in range(hp.num_layers - 1):
        cre = cre_layer(cre, ctx, cu_c, sd, f"{pfx}cre_layers.{i}.", hp, slopes, rnd, last=i == nl - 2,
                        ctx_labels=labels if hp.use_context else None)
        if collect is not None and i == 0:
            collect["first_cre_layer_out"] = cre.repeat(T, 1)
        gene = gene_layer(gene, cre, f"{pfx}gene_layers.{i + 1}.", i + 1 == nl - 1)
    return gene, cre


def predict_step(batch: dict, sd: dict, cre_hp, gene_hp, hp, rounding=None, share_cre_stream=False):
    """Seq2GenePredictorCombinedModulator.predict_step (model_combined_modulator.py:857-907):
    per-gene split of the fp32 outputs."""
    with torch.no_grad():
        pred, emb = forward(batch, sd, cre_hp, gene_hp, hp, rounding, share_cre_stream)
    preds, embs, s = [], [], 0
    for t in batch["tissue_context"]:
        n = len(t)
        preds.append(pred[s:s + n].numpy())
        embs.append(emb[s:s + n].numpy())
        s += n
    return {"pred_gene_exp": preds, "embeddings": embs, "batch_idx": 0, "dataloader_idx": None}


def variant_prediction(batch: dict, sd: dict, cre_hp, gene_hp, hp, rounding=None, share_cre_stream=False):
    """Seq2GenePredictorCombinedModulator.variant_prediction (model_combined_modulator.py:909-1004): the ref / het / hom
    samples go through forward one at a time (:944-962) with their token positions; the modulator returns the final
    gene-stream row at gene_token_position (+1 when a start / registry token is prepended, :665-666) and the row of the
    CRE stream after its last layer at cre_token_position, repeated for every requested tissue (:296-326, :631-639).
    NaN positions (variant outside every CRE / chunk) switch the gather off: zeros (:936-939, :303-306, :322-325)."""
    n = len(batch["cre_sequences"])
    cre_pos, gene_pos = batch["cre_token_position"], batch["gene_token_position"]
    assert len(cre_pos) == 3 and len(gene_pos) == 3
    if torch.isnan(torch.as_tensor(cre_pos, dtype=torch.float32)).any():
        cre_pos = None
    if torch.isnan(torch.as_tensor(gene_pos, dtype=torch.float32)).any():
        gene_pos = None
    prefix = 0 if hp.gene_pooling == "max" else 1
    out = {"pred_gene_exp": [], "embd": [], "variant_type": batch["variant_type"], "gene_token_embedding": [],
           "cre_token_embedding": []}
    for i in range(n):
        one = {"cre_sequences": batch["cre_sequences"][i:i + 1], "cre_attention_masks": batch["cre_attention_masks"][i:i + 1],
               "tissue_context": batch["tissue_context"][i:i + 1], "ref_cre_labels": batch["ref_labels"][i:i + 1],
               "gene_embeddings": batch["gene_embeddings"][i:i + 1], "gene_attention_masks": batch["gene_attention_masks"][i:i + 1]}
        col = {}
        with torch.no_grad():
            pred, emb = forward(one, sd, cre_hp, gene_hp, hp, rounding, share_cre_stream, collect=col)
        T = pred.shape[0]
        g_out, c_out = col["modulator_gene_out"][0], col["modulator_cre_out"][0]
        gt = torch.zeros(T, hp.emb_dim) if gene_pos is None else \
            g_out[:, int(torch.as_tensor(gene_pos[i]).reshape(-1)[0]) + prefix, :]
        ct = torch.zeros(T, hp.emb_dim) if cre_pos is None else c_out[:, int(torch.as_tensor(cre_pos[i]).reshape(-1)[0]), :]
        out["pred_gene_exp"].append(pred.numpy())
        out["embd"].append(emb.numpy())
        out["gene_token_embedding"].append(gt.numpy().copy())
        out["cre_token_embedding"].append(ct.numpy().copy())
    return out
