"""Tensor-level wrappers over the C ABI of libvf_hip.so.

torch is used only as the owner of device memory and of the HIP stream; every computation is a
hand-written HIP kernel.  All wrappers require CUDA(ROCm) tensors and raise otherwise.
"""
from __future__ import annotations

import math

import torch

from . import _lib, runtime
from ._lib import (EPI_BF16, EPI_F32, EPI_GEGLU_BF16, EPI_GELU_BF16, EPI_GELU_F32, EPI_RES_F32, VF_BF16,  # noqa: F401
                   VF_F16, VF_F32, check)

# The 16-bit operand type of the GEMM / attention kernels: torch.bfloat16 (the reference's shipped `bf16-mixed`) or
# torch.float16 (its `16-mixed` / fp16 flash-attn path; BASELINE configs[4]).  fp32 accumulation either way.  The
# "*_BF16" epilogue names mean "16-bit output in the operand type".
_CDT = torch.bfloat16


class compute_dtype:
    """with ops.compute_dtype(torch.float16): ...  -- 16-bit tensors created inside (LayerNorm outputs, casts, packed
    weights) use this operand type; kernels themselves follow the dtype of the tensors they are handed."""

    def __init__(self, dtype):
        assert dtype in (torch.bfloat16, torch.float16), dtype
        self.dtype = dtype

    def __enter__(self):
        global _CDT
        self.prev, _CDT = _CDT, self.dtype

    def __exit__(self, *exc):
        global _CDT
        _CDT = self.prev
        return False


def cdt():
    return _CDT


def _is16(dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Brackets individual kernel launches with HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON (torch's
    current stream, which is the stream handed to the C ABI).  Used by bench.py for the roofline numbers."""

    def __init__(self, detail: bool = False):
        self.records = {}
        self.detail = detail          # also key records by launch geometry (scripts/shape_breakdown.py)
        self.order = []               # (name, geometry, family, flops | None, bytes) of every launch, in launch order
                                      # (scripts/pmc_shapes.py joins it with the dispatch order of a rocprofv3 counter pass)

    def time(self, name: str, flops, nbytes: float, launch, geometry: str = "", family: str = ""):
        """flops: a number, or a zero-argument callable evaluated in summary() (attention: the per-sequence lengths
        live on the device, and reading them here would put a host sync between the launches being timed)."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = launch()
        b.record()
        rec = [a, b, flops, nbytes]
        self.records.setdefault(name, []).append(rec)
        if self.detail:
            self.order.append((name, geometry, family, None if callable(flops) else float(flops), float(nbytes)))
        if family:
            self.records.setdefault(f"{name}:{family}", []).append(rec)
        if self.detail and geometry:
            self.records.setdefault(f"{name}[{geometry}]", []).append(rec)
        return out

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            for r in recs:
                if callable(r[2]):
                    r[2] = float(r[2]())
            ms = sum(a.elapsed_time(b) for a, b, _, _ in recs)
            out[name] = {"launches": len(recs), "total_ms": ms, "flops": sum(r[2] for r in recs),
                         "bytes": sum(r[3] for r in recs)}
        return out


TIMER: KernelTimer | None = None
_SCOPE = ""          # default family label of launches made inside a `scope(...)` block (timer bookkeeping only)


class scope:
    """with ops.scope("seq2reg"): ...  -- launches without an explicit family are recorded under this label."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        global _SCOPE
        self.prev, _SCOPE = _SCOPE, self.name

    def __exit__(self, *exc):
        global _SCOPE
        _SCOPE = self.prev
        return False


def _dev(*ts):
    """Every tensor must live on the CURRENT device: the C ABI launches on torch's current stream and never switches
    devices (one process per GPU; use torch.cuda.set_device / torch.cuda.device(...) around calls otherwise)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.VFError("variantformer_amd ops need tensors on the GPU (no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise _lib.VFError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                               "libvf_hip launches on the current device's stream (wrap the call in torch.cuda.device)")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _dt(dtype) -> int:
    if dtype == torch.float32:
        return VF_F32
    if dtype == torch.bfloat16:
        return VF_BF16
    if dtype == torch.float16:
        return VF_F16
    raise _lib.VFError(f"unsupported dtype {dtype}")


def gemm(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, epilogue: int,
         residual: torch.Tensor | None = None, out: torch.Tensor | None = None, variant: int = 0,
         family: str = "") -> torch.Tensor:
    """out = epilogue(a[M,K] @ w[N,K]^T + bias).  a, w both bf16 or both fp16 (a may be a row-strided view); 16-bit
    outputs come back in the operand type.  variant != 0 forces a tile configuration (vf_gemm_*_ex; tests only)."""
    _dev(a, w, bias, residual, out)
    assert _is16(a.dtype) and w.dtype == a.dtype and a.dim() == 2 and w.dim() == 2
    assert a.stride(1) == 1 and w.is_contiguous() and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue == EPI_GEGLU_BF16 else N
    odt = torch.float32 if epilogue in (EPI_F32, EPI_RES_F32, EPI_GELU_F32) else a.dtype
    if out is None:
        out = torch.empty((M, n_out), dtype=odt, device=a.device)
    assert out.dtype == odt and out.shape == (M, n_out) and out.stride(1) == 1
    ldr = 0
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    lib = _lib.load()

    def launch():
        lda = a.stride(0) if M > 1 else max(a.stride(0), K)
        ldo = out.stride(0) if M > 1 else max(out.stride(0), n_out)
        f16 = a.dtype == torch.float16
        if variant:
            fn = lib.vf_gemm_f16_ex if f16 else lib.vf_gemm_bf16_ex
            check(fn(a.data_ptr(), lda, w.data_ptr(), _ptr(bias), _ptr(residual), ldr, out.data_ptr(),
                     ldo, M, N, K, epilogue, variant, _stream()), "vf_gemm_ex")
        else:
            fn = lib.vf_gemm_f16 if f16 else lib.vf_gemm_bf16
            check(fn(a.data_ptr(), lda, w.data_ptr(), _ptr(bias), _ptr(residual), ldr, out.data_ptr(),
                     ldo, M, N, K, epilogue, _stream()), "vf_gemm")
    if TIMER is not None:
        nbytes = 2.0 * (M * K + N * K) + out.numel() * out.element_size() + (0 if residual is None else 4.0 * M * N)
        TIMER.time("gemm", 2.0 * M * N * K, nbytes, launch, f"M={M} N={N} K={K} epi={epilogue}", family or _SCOPE)
    else:
        launch()
    return out


class LnStream:
    """A residual stream as the LayerNorm-folding GEMMs exchange it: x fp32 [M, D] (None when the fp32 rows have no
    reader), x16 its 16-bit copy -- for fp16 streams stored SCALED, x16 = fp16(x * scale) --, stats fp32 [M, 2] =
    (mean * scale, rstd / scale) of every row: exactly the pair the consumer GEMM applies to its accumulator of the
    scaled operand (vf_gemm_ln / vf_ln_finalize2 / vf_row_stats_cast2).  scale = 1 for bf16 streams."""
    __slots__ = ("x", "x16", "stats", "scale", "t16")

    def __init__(self, x, x16, stats, scale: float = 1.0, t16=None):
        self.x, self.x16, self.stats, self.scale = x, x16, stats, float(scale)
        # fp16 trunk copy fp16(x * T16_SCALE) (layers.trunk16_mode() == "f16"): what the NEXT layer's down-projection adds
        # as its residual when the fp32 rows are not stored (vf_gemm_ln_t16)
        self.t16 = t16

    def operand16(self):
        """The stream as a plain 16-bit GEMM operand (un-normalised use: the K/V projection of a cross attention):
        the copy itself when it is unscaled, else a cast of the fp32 rows."""
        return self.x16 if self.scale == 1.0 else cast16(self.x, self.x16.dtype)


def x16_scale_for(dtype) -> float:
    """Power-of-two scale of the 16-bit copy of a residual stream: 1 for bf16 (fp32's exponent range); 2^-4 for fp16, which
    keeps |x| < 1e6 representable (the reference's own fp16 autocast overflows at 65504) at no cost in precision --
    10 mantissa bits at every magnitude above 1e-3 -- and LayerNorm is scale-invariant."""
    return 1.0 if dtype != torch.float16 else 2.0 ** -4


# Rows whose |mean| exceeds this many standard deviations make the folded LayerNorm -> Linear lose accuracy (it rounds
# the UNCENTRED row to 16 bits; tests/test_ops_gpu.py::test_ln_fold_rows_with_large_mean: error ~ 2^-9 x |mean| / std).
# The statistics kernels raise a device flag; ln_fold_alert() reads it (one int, with the outputs of a batch).
LN_FOLD_RATIO_LIMIT = 8.0
_ALERT: dict = {}


def _alert_flag(device) -> torch.Tensor:
    """The flag of the CURRENT STREAM of `device`: kernels raise the flag of the stream they run on, so concurrent streams
    (one model per stream / thread) never see each other's rows, and ln_fold_alert_take() -- enqueued on the same stream
    right after a batch's kernels -- cuts out exactly that batch's bits."""
    sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, sid)
    if key not in _ALERT:
        _ALERT[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _ALERT[key]


def ln_fold_alert_clear(device) -> None:
    """Enqueue a reset of the current stream's flag (start of a batch: bits left by direct forward() calls, benchmarks or a
    discarded batch belong to nobody)."""
    if device.type == "cuda":
        _alert_flag(device).zero_()


def ln_fold_alert_take(device):
    """Enqueue `bits = flag; flag = 0` on the current stream and return the device tensor `bits` ([1] int32): the alert of
    everything this stream ran since the last clear / take, i.e. of ONE batch when called where the batch's last kernel was
    enqueued.  Does not synchronise; read it with the batch's outputs.  None on a CPU device."""
    if device.type != "cuda":
        return None
    flag = _alert_flag(device)
    bits = flag.clone()
    flag.zero_()
    return bits


# Elements at or beyond this magnitude could overflow a scaled fp16 copy of a stream (65504 / 2^-4 with some margin): the
# statistics kernels bound every element of a row by |mean| + sqrt(D * var) and raise bit 1 of the flag (ABI 6).
def ln_fold_abs_limit() -> float:
    """Largest |x| the 16-bit copies of a LayerNorm-folded stream represent: unbounded (0 = no check) when every copy is
    bf16; 60000 / scale for the fp16 copy with the least headroom (fp16 operand copy, fp16 trunk copy)."""
    scales = []
    if _CDT == torch.float16:
        scales.append(x16_scale_for(torch.float16))
    if runtime.env().trunk16 == "f16":           # VF_TRUNK16, read once per forward (runtime.forward_env)
        scales.append(T16_SCALE)
    return 60000.0 / max(scales) if scales else 0.0


def ln_fold_alert(device, reset: bool = True) -> int:
    """Non-zero when, since the last reset, a LayerNorm-folded stream ON THE CURRENT STREAM left the regime its 16-bit copies
    serve: bit 0 -- some row had |mean| > LN_FOLD_RATIO_LIMIT standard deviations; bit 1 -- some row may hold an element
    beyond ln_fold_abs_limit() (a scaled fp16 copy could overflow).  Synchronises (op-level callers and tests; the model
    carries a per-batch copy instead: ln_fold_alert_take).  The model recomputes a flagged batch with the separate LayerNorm
    on fp32 rows (layers.ln_fold_forced_off) -- degraded numbers are never returned."""
    if device.type != "cuda":
        return 0
    sid = torch.cuda.current_stream(device).cuda_stream
    key = (device.type, device.index, sid)
    if key not in _ALERT:
        return 0
    hit = int(_ALERT[key].item())
    if hit and reset:
        _ALERT[key].zero_()
    return hit


def ln_stream(x: torch.Tensor, eps: float = 1e-5, raise_alert: bool = True) -> LnStream:
    """(x, 16-bit copy, row statistics) for a stream no GEMM produced: one pass over x (vf_row_stats_cast2).
    raise_alert=False: the rows are a TABLE of which a batch uses only some (the registry table in front of the first gene
    layer); the caller raises the alert for the rows in use itself."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    M, D = x.shape
    x16 = torch.empty((M, D), dtype=_CDT, device=x.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    scale = x16_scale_for(_CDT)
    alert_ptr = _alert_flag(x.device).data_ptr() if raise_alert else 0

    def launch():
        check(_lib.load().vf_row_stats_cast2(x.data_ptr(), M, D, eps, x16.data_ptr(), _dt(_CDT), scale, LN_FOLD_RATIO_LIMIT,
                                             ln_fold_abs_limit(), alert_ptr, stats.data_ptr(), _stream()),
              "vf_row_stats_cast")
    if TIMER is not None:
        TIMER.time("layernorm", 0.0, float(M) * D * 6, launch, f"stats_cast D={D}", _SCOPE)
    else:
        launch()
    return LnStream(x, x16, stats, scale)


def ln_stream_rows(s: LnStream, rows: torch.Tensor) -> LnStream:
    """The rows `rows` (int64) of a stream, statistics included (bit-identical to the full stream's)."""
    return LnStream(None if s.x is None else gather_rows_f32(s.x, None, rows), gather_rows_bf16(s.x16, rows),
                    gather_rows_f32(s.stats, None, rows), s.scale,
                    None if s.t16 is None else gather_rows_bf16(s.t16, rows))


T16_SCALE = 2.0 ** -4          # the fp16 trunk copy is fp16(x * 2^-4): |x| < 1e6 representable (x16_scale_for(fp16))


def trunk16_of(x: torch.Tensor, raise_alert: bool = True) -> torch.Tensor:
    """fp16(x * T16_SCALE) of an fp32 stream no trunk-writing GEMM produced (a stack's input): one pass
    (vf_row_stats_cast2 with an fp16 output; its statistics are not used).  raise_alert=False: a table of which a batch uses
    only some rows (see ln_stream)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    M, D = x.shape
    t16 = torch.empty((M, D), dtype=torch.float16, device=x.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)

    alert_ptr = _alert_flag(x.device).data_ptr() if raise_alert else 0

    def launch():       # ratio limit off (1e30): this stream's statistics are the caller's business; range check on
        check(_lib.load().vf_row_stats_cast2(x.data_ptr(), M, D, 1e-5, t16.data_ptr(), VF_F16, T16_SCALE, 1e30,
                                             60000.0 / T16_SCALE, alert_ptr, stats.data_ptr(), _stream()),
              "vf_row_stats_cast")
    if TIMER is not None:
        TIMER.time("layernorm", 0.0, float(M) * D * 6, launch, f"trunk16_of D={D}", _SCOPE)
    else:
        launch()
    return t16


def gemm_ln_consumer(s: LnStream, w: torch.Tensor, bias: torch.Tensor, colsum: torch.Tensor, epilogue: int,
                     family: str = "") -> torch.Tensor:
    """epilogue(LN(s.x) @ W^T + b) without materialising LN(s.x): w = 16-bit(gamma (.) W) [N, K], bias = W beta + b,
    colsum = rowsum(w) (layers.packed_linear_ln).  epilogue EPI_BF16 or EPI_GEGLU_BF16 (16-bit output, operand type), or
    EPI_F32 (fp32 rows: the logits of softmax_counted)."""
    a = s.x16
    _dev(a, w, bias, colsum, s.stats)
    assert _is16(a.dtype) and w.dtype == a.dtype and a.shape[1] == w.shape[1] and a.stride(1) == 1
    assert epilogue in (EPI_BF16, EPI_GEGLU_BF16, EPI_F32)
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue == EPI_GEGLU_BF16 else N
    out = torch.empty((M, n_out), dtype=torch.float32 if epilogue == EPI_F32 else a.dtype, device=a.device)
    lib = _lib.load()

    def launch():
        check(lib.vf_gemm_ln(a.data_ptr(), a.stride(0) if M > 1 else max(a.stride(0), K), w.data_ptr(), bias.data_ptr(),
                             0, 0, VF_F32, out.data_ptr(), n_out, M, N, K, epilogue, _dt(a.dtype), s.stats.data_ptr(),
                             colsum.data_ptr(), 0, 0, 0, 1.0, 1.0, _stream()), "vf_gemm_ln")
    if TIMER is not None:
        nbytes = 2.0 * (M * K + N * K) + out.numel() * out.element_size() + 8.0 * M
        TIMER.time("gemm", 2.0 * M * N * K, nbytes, launch, f"M={M} N={N} K={K} epi={epilogue} ln=consumer", family or _SCOPE)
    else:
        launch()
    return out


def gemm_ln_producer(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, residual,
                     eps: float = 1e-5, family: str = "", need_x: bool = True, trunk16: torch.Tensor | None = None,
                     need_t16: bool = False) -> LnStream:
    """x = a @ w^T + bias (+ residual), fp32, returned together with its 16-bit copy and row statistics (the next
    LayerNorm's): EPI_RES_F32 when a residual is given, else EPI_F32.  `residual`: an fp32 tensor [M, N], or an LnStream
    whose 16-BIT COPY is the residual (its fp32 rows need not exist).  need_x=False: the fp32 x itself has no reader (a
    layer's intermediate stream feeds only the next LayerNorm -> Linear pair) and is not stored; .x is None.
    `trunk16` (instead of `residual`): the residual as the fp16 trunk copy fp16(r * T16_SCALE) of the layer input
    (vf_gemm_ln_t16); need_t16: also write that copy of the result (.t16) for the next layer."""
    if trunk16 is not None:
        assert residual is None
        return _gemm_ln_producer_t16(a, w, bias, trunk16, eps, family, need_x, need_t16)
    assert not need_t16
    res16 = residual if isinstance(residual, LnStream) else None
    res32 = None if res16 is not None else residual
    _dev(a, w, bias, res32, None if res16 is None else res16.x16)
    assert _is16(a.dtype) and w.dtype == a.dtype and a.shape[1] == w.shape[1] and a.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device) if need_x else None
    x16 = torch.empty((M, N), dtype=a.dtype, device=a.device)
    n_parts = (N + 31) // 32
    part = torch.empty((n_parts, M, 2), dtype=torch.float32, device=a.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=a.device)
    epi = EPI_F32 if residual is None else EPI_RES_F32
    if res32 is not None:
        assert res32.dtype == torch.float32 and res32.shape == (M, N) and res32.stride(1) == 1
        r_ptr, r_ld, r_dt, r_scale = res32.data_ptr(), res32.stride(0), VF_F32, 1.0
    elif res16 is not None:
        r = res16.x16
        assert r.dtype == a.dtype and r.shape == (M, N) and r.stride(1) == 1
        r_ptr, r_ld, r_dt, r_scale = r.data_ptr(), r.stride(0), _dt(r.dtype), 1.0 / res16.scale
    else:
        r_ptr, r_ld, r_dt, r_scale = 0, 0, VF_F32, 1.0
    scale = x16_scale_for(a.dtype)
    alert = _alert_flag(a.device)
    lib = _lib.load()

    def launch():
        check(lib.vf_gemm_ln(a.data_ptr(), a.stride(0) if M > 1 else max(a.stride(0), K), w.data_ptr(), _ptr(bias),
                             r_ptr, r_ld, r_dt, _ptr(out), N, M, N, K, epi, _dt(a.dtype), 0, 0, x16.data_ptr(), N,
                             part.data_ptr(), scale, r_scale, _stream()), "vf_gemm_ln")

    def finalize():
        check(lib.vf_ln_finalize2(part.data_ptr(), M, n_parts, N, eps, scale, LN_FOLD_RATIO_LIMIT, ln_fold_abs_limit(),
                                  alert.data_ptr(), stats.data_ptr(), _stream()), "vf_ln_finalize")
    if TIMER is not None:
        res_bytes = 0.0 if residual is None else (4.0 if res32 is not None else 2.0) * M * N
        nbytes = 2.0 * (M * K + N * K) + M * N * ((4.0 if need_x else 0.0) + 2.0) + res_bytes + 8.0 * M * n_parts
        tag = "producer" + ("" if need_x else "-nox") + ("-r16" if res16 is not None else "")
        TIMER.time("gemm", 2.0 * M * N * K, nbytes, launch, f"M={M} N={N} K={K} epi={epi} ln={tag}", family or _SCOPE)
        TIMER.time("layernorm", 0.0, 8.0 * M * (n_parts + 1), finalize, f"finalize D={N}", _SCOPE)
    else:
        launch()
        finalize()
    return LnStream(out, x16, stats, scale)


def _gemm_ln_producer_t16(a, w, bias, t_in, eps, family, need_x, need_t16) -> LnStream:
    _dev(a, w, bias, t_in)
    assert _is16(a.dtype) and w.dtype == a.dtype and a.shape[1] == w.shape[1] and a.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert t_in.dtype == torch.float16 and t_in.shape == (M, N) and t_in.stride(1) == 1
    out = torch.empty((M, N), dtype=torch.float32, device=a.device) if need_x else None
    t_out = torch.empty((M, N), dtype=torch.float16, device=a.device) if need_t16 else None
    x16 = torch.empty((M, N), dtype=a.dtype, device=a.device)
    n_parts = (N + 31) // 32
    part = torch.empty((n_parts, M, 2), dtype=torch.float32, device=a.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=a.device)
    scale = x16_scale_for(a.dtype)
    alert = _alert_flag(a.device)
    lib = _lib.load()

    def launch():
        check(lib.vf_gemm_ln_t16(a.data_ptr(), a.stride(0) if M > 1 else max(a.stride(0), K), w.data_ptr(), _ptr(bias),
                                 t_in.data_ptr(), t_in.stride(0), 1.0 / T16_SCALE, _ptr(out), N, M, N, K, _dt(a.dtype),
                                 x16.data_ptr(), N, part.data_ptr(), scale, _ptr(t_out), N, T16_SCALE, _stream()),
              "vf_gemm_ln_t16")

    def finalize():
        check(lib.vf_ln_finalize2(part.data_ptr(), M, n_parts, N, eps, scale, LN_FOLD_RATIO_LIMIT, ln_fold_abs_limit(),
                                  alert.data_ptr(), stats.data_ptr(), _stream()), "vf_ln_finalize")
    if TIMER is not None:
        nbytes = 2.0 * (M * K + N * K) + M * N * ((4.0 if need_x else 0.0) + (2.0 if need_t16 else 0.0) + 2.0 + 2.0) + 8.0 * M * n_parts
        tag = "producer" + ("" if need_x else "-nox") + "-t16" + ("" if need_t16 else "-in")
        TIMER.time("gemm", 2.0 * M * N * K, nbytes, launch, f"M={M} N={N} K={K} epi={EPI_RES_F32} ln={tag}", family or _SCOPE)
        TIMER.time("layernorm", 0.0, 8.0 * M * (n_parts + 1), finalize, f"finalize D={N}", _SCOPE)
    else:
        launch()
        finalize()
    return LnStream(out, x16, stats, scale, t_out)


def pack_geglu_rows(w: torch.Tensor, bias: torch.Tensor | None):
    """Permute a [2F,K] 16-bit weight (+ fp32 bias) into the GEGLU-epilogue row order."""
    _dev(w, bias)
    assert _is16(w.dtype) and w.is_contiguous()
    wo = torch.empty_like(w)
    bo = torch.empty_like(bias) if bias is not None else None
    check(_lib.load().vf_pack_geglu_rows(w.data_ptr(), _ptr(bias), wo.data_ptr(), _ptr(bo), w.shape[0], w.shape[1],
                                         _stream()), "vf_pack_geglu_rows")
    return wo, bo


ATTN_Q_AT_START, ATTN_Q_LOG2 = 1, 2          # enum vf_attn_flags


def attn_counted_keys(q: torch.Tensor, kv_table: torch.Tensor, log2_count: torch.Tensor, cu_q: torch.Tensor, max_q: int,
                      n_heads: int, head_dim: int, family: str = "") -> torch.Tensor:
    """Cross attention against keys that are copies of the C <= 16 distinct rows of kv_table [C, 2 H dh] (K | V), row c
    occurring 2^log2_count[s, c] times among sequence s's keys (vf_attn_counted_keys): softmax over the distinct rows with
    log2(count) added to the base-2 logit.  q [tokens, >= H dh] 16-bit, pre-scaled (the model's q_log2 form)."""
    _dev(q, kv_table, log2_count, cu_q)
    assert _is16(q.dtype) and kv_table.dtype == q.dtype and q.stride(1) == 1 and kv_table.stride(1) == 1
    assert log2_count.dtype == torch.float32 and log2_count.is_contiguous() and cu_q.dtype == torch.int32
    D = n_heads * head_dim
    n_seq, C = cu_q.numel() - 1, kv_table.shape[0]
    assert log2_count.shape == (n_seq, C) and kv_table.shape[1] >= 2 * D
    out = torch.empty((q.shape[0], D), dtype=q.dtype, device=q.device)

    def launch():
        check(_lib.load().vf_attn_counted_keys(q.data_ptr(), q.stride(0), kv_table.data_ptr(), kv_table.stride(0),
                                               log2_count.data_ptr(), cu_q.data_ptr(), n_seq, int(max_q), C, n_heads, head_dim,
                                               out.data_ptr(), out.stride(0), _dt(q.dtype), _stream()), "vf_attn_counted_keys")
    if TIMER is not None:       # executed work: 4 * tokens * C * width
        TIMER.time("attn", 4.0 * q.shape[0] * C * D, 2.0 * D * 2 * q.shape[0], launch,
                   f"H={n_heads} dh={head_dim} max_q={int(max_q)} counted_keys={C}", family or _SCOPE)
    else:
        launch()
    return out


def softmax_counted(scores: torch.Tensor, log2_count: torch.Tensor, cu_q: torch.Tensor, max_q: int, n_heads: int, slots: int,
                    out_dtype=None, family: str = "") -> torch.Tensor:
    """16-bit softmax over the C = log2_count.shape[1] distinct keys of every (token, head), log2(count) added to the base-2
    logits scores fp32 [tokens, H * slots] (slots >= C per head, the padding gets weight 0): vf_softmax_counted."""
    _dev(scores, log2_count, cu_q)
    assert scores.dtype == torch.float32 and scores.stride(1) == 1 and log2_count.dtype == torch.float32 and log2_count.is_contiguous()
    T, n_seq, C = scores.shape[0], cu_q.numel() - 1, log2_count.shape[1]
    assert scores.shape[1] >= n_heads * slots and log2_count.shape[0] == n_seq
    dt = _CDT if out_dtype is None else out_dtype
    out = torch.empty((T, n_heads * slots), dtype=dt, device=scores.device)

    def launch():
        check(_lib.load().vf_softmax_counted(scores.data_ptr(), scores.stride(0), log2_count.data_ptr(), cu_q.data_ptr(), n_seq,
                                             int(max_q), n_heads, slots, C, out.data_ptr(), out.stride(0), _dt(dt), _stream()),
              "vf_softmax_counted")
    if TIMER is not None:
        TIMER.time("attn", 0.0, float(T) * n_heads * slots * 6, launch, f"softmax_counted H={n_heads} C={C}", family or _SCOPE)
    else:
        launch()
    return out


def attn_rows_supported(head_dim: int, alibi: bool, n_seq: int, n_heads: int, max_q: int, max_k: int, q_log2: bool) -> bool:
    """Whether attn_varlen(rows=...) has a kernel for this geometry (vf_attn_rows_supported); otherwise the caller gathers the
    rows (gather_rows_bf16) and calls the plain form -- same bits either way."""
    return bool(_lib.load().vf_attn_rows_supported(int(head_dim), int(bool(alibi)), int(n_seq), int(n_heads), int(max_q),
                                                   int(max_k), ATTN_Q_LOG2 if q_log2 else 0))


def attn_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_q: torch.Tensor, cu_k: torch.Tensor | None,
                max_q: int, max_k: int, n_heads: int, head_dim: int, slopes: torch.Tensor | None = None,
                scale: float | None = None, out: torch.Tensor | None = None, q_at_start: bool = False,
                family: str = "", q_log2: bool = False, rows: torch.Tensor | None = None) -> torch.Tensor:
    """q [tq, >=H*dh] / k, v [tk, >=H*dh] bf16 row-strided views whose first H*dh columns are the heads.
    q_at_start: ALiBi positions of the queries count from the start of the key sequence (default: flash-attn's
    end alignment).  q_log2: q was projected with weights pre-multiplied by scale * log2(e) (layers.q_prescale): q . k is the
    base-2 logit, `scale` is not applied again (VF_ATTN_Q_LOG2).
    rows int64 [tokens] (self attention only, vf_attn_varlen_fwd_rows): q / k / v are tables of distinct rows, token t's
    row is rows[t]; the gather happens in the kernel's loads (attn_rows_supported says for which geometries)."""
    _dev(q, k, v, cu_q, cu_k, slopes, out, rows)
    for t in (q, k, v):
        assert _is16(t.dtype) and t.dtype == q.dtype and t.dim() == 2 and t.stride(1) == 1
    assert cu_q.dtype == torch.int32 and (cu_k is None or cu_k.dtype == torch.int32)
    D = n_heads * head_dim
    tq = q.shape[0] if rows is None else rows.numel()
    tk = k.shape[0] if rows is None else rows.numel()
    if rows is not None:
        assert rows.dtype == torch.int64 and rows.is_contiguous() and cu_k is None
    if out is None:
        out = torch.empty((tq, D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    if slopes is not None:
        assert slopes.dtype == torch.float32 and slopes.numel() == n_heads
    def launch():
        lib = _lib.load()
        flags = (ATTN_Q_AT_START if q_at_start else 0) | (ATTN_Q_LOG2 if q_log2 else 0)
        if rows is not None:
            check(lib.vf_attn_varlen_fwd_rows(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(0), k.stride(0),
                                              v.stride(0), out.stride(0), cu_q.data_ptr(), None, cu_q.numel() - 1, int(max_q),
                                              int(max_k), n_heads, head_dim, _ptr(slopes), float(scale), _dt(q.dtype), flags,
                                              rows.data_ptr(), rows.data_ptr(), _stream()), "vf_attn_varlen_fwd_rows")
            return
        check(lib.vf_attn_varlen_fwd_v2(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(0), k.stride(0),
                                        v.stride(0), out.stride(0), cu_q.data_ptr(), _ptr(cu_k), cu_q.numel() - 1, int(max_q),
                                        int(max_k), n_heads, head_dim, _ptr(slopes), float(scale), _dt(q.dtype), flags,
                                        _stream()), "vf_attn_varlen_fwd")
    if TIMER is not None:
        def flops():       # 4 * sum_seq(len_q * len_k) * H * dh (QK^T and PV), evaluated after the timed replay
            lq = (cu_q[1:] - cu_q[:-1]).double()
            lk = lq if cu_k is None else (cu_k[1:] - cu_k[:-1]).double()
            return 4.0 * float((lq * lk).sum().item()) * D
        TIMER.time("attn", flops, 2.0 * D * (2 * tq + 2 * tk), launch,
                   f"H={n_heads} dh={head_dim} max_q={int(max_q)} max_k={int(max_k)}" + (" rows" if rows is not None else ""),
                   family or _SCOPE)
    else:
        launch()
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype=None, gelu: bool = False,
              eps: float = 1e-5, out: torch.Tensor | None = None) -> torch.Tensor:
    """out_dtype None = the current compute dtype (ops.cdt())."""
    _dev(x, gamma, beta, out)
    out_dtype = _CDT if out_dtype is None else out_dtype
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    rows, D = x.shape
    if out is None:
        out = torch.empty((rows, D), dtype=out_dtype, device=x.device)

    def launch():
        check(_lib.load().vf_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), rows, D, eps,
                                       _dt(out.dtype), int(gelu), _stream()), "vf_layernorm")
    if TIMER is not None:
        TIMER.time("layernorm", 0.0, float(rows) * D * (4 + out.element_size()), launch, f"D={D}", _SCOPE)
    else:
        launch()
    return out


def mask_to_cu_seqlens(pad: torch.Tensor) -> torch.Tensor:
    """pad bool/uint8 [W, L] (True = pad) -> int32 [W+1] exclusive prefix sum of valid counts."""
    _dev(pad)
    pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad
    assert pad.dtype == torch.uint8 and pad.is_contiguous() and pad.dim() == 2
    W, L = pad.shape
    cu = torch.empty(W + 1, dtype=torch.int32, device=pad.device)
    check(_lib.load().vf_mask_to_cu_seqlens(pad.data_ptr(), cu.data_ptr(), W, L, _stream()), "vf_mask_to_cu_seqlens")
    return cu


def embed_pack(ids: torch.Tensor, pad: torch.Tensor, cu: torch.Tensor, table: torch.Tensor,
               pos_table: torch.Tensor | None, n_tokens: int) -> torch.Tensor:
    _dev(ids, pad, cu, table, pos_table)
    pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2 and pad.shape == ids.shape
    assert table.dtype == torch.float32 and table.is_contiguous()
    W, L = ids.shape
    d = table.shape[1]
    out = torch.empty((n_tokens, d), dtype=torch.float32, device=ids.device)
    check(_lib.load().vf_embed_pack(ids.data_ptr(), pad.data_ptr(), cu.data_ptr(), table.data_ptr(), _ptr(pos_table),
                                    out.data_ptr(), W, L, d, table.shape[0], _stream()), "vf_embed_pack")
    return out


def embed_stream(ids: torch.Tensor, pad: torch.Tensor, cu: torch.Tensor, table: torch.Tensor,
                 pos_table: torch.Tensor | None, n_tokens: int, need_x: bool = False, need_t16: bool = True,
                 eps: float = 1e-5) -> LnStream:
    """embed_pack + ln_stream (+ trunk16_of) in one kernel (vf_embed_stream): the encoder input as an LnStream whose fp32
    rows exist only on request; bit-identical copies and statistics to the three-step form."""
    _dev(ids, pad, cu, table, pos_table)
    pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2 and pad.shape == ids.shape
    assert table.dtype == torch.float32 and table.is_contiguous()
    W, L = ids.shape
    d = table.shape[1]
    dev = ids.device
    x = torch.empty((n_tokens, d), dtype=torch.float32, device=dev) if need_x else None
    x16 = torch.empty((n_tokens, d), dtype=_CDT, device=dev)
    t16 = torch.empty((n_tokens, d), dtype=torch.float16, device=dev) if need_t16 else None
    stats = torch.empty((n_tokens, 2), dtype=torch.float32, device=dev)
    scale = x16_scale_for(_CDT)
    alert = _alert_flag(dev)

    def launch():
        check(_lib.load().vf_embed_stream(ids.data_ptr(), pad.data_ptr(), cu.data_ptr(), table.data_ptr(), _ptr(pos_table),
                                          _ptr(x), x16.data_ptr(), _dt(_CDT), scale, _ptr(t16), T16_SCALE, stats.data_ptr(),
                                          eps, LN_FOLD_RATIO_LIMIT, ln_fold_abs_limit(), alert.data_ptr(), W, L, d, table.shape[0],
                                          _stream()),
              "vf_embed_stream")
    if TIMER is not None:
        TIMER.time("layernorm", 0.0, float(n_tokens) * d * (2 + (2 if need_t16 else 0) + (4 if need_x else 0)), launch,
                   f"embed_stream D={d}", _SCOPE)
    else:
        launch()
    return LnStream(x, x16, stats, scale, t16)


def token_keys(ids: torch.Tensor, pad: torch.Tensor, cu: torch.Tensor, n_tokens: int, vocab: int, key_L: int) -> torch.Tensor:
    """int64 [n_tokens]: id * key_L + position of every packed valid token (vf_token_keys; key_L = 1: the ids themselves)."""
    _dev(ids, pad, cu)
    pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2 and pad.shape == ids.shape and pad.is_contiguous()
    W, L = ids.shape
    keys = torch.empty((n_tokens,), dtype=torch.int64, device=ids.device)
    check(_lib.load().vf_token_keys(ids.data_ptr(), pad.data_ptr(), cu.data_ptr(), keys.data_ptr(), W, L, int(vocab), int(key_L),
                                    _stream()), "vf_token_keys")
    return keys


def segment_mean(x: torch.Tensor, cu: torch.Tensor, out_dtype=None) -> torch.Tensor:
    _dev(x, cu)
    out_dtype = _CDT if out_dtype is None else out_dtype
    assert x.dtype == torch.float32 and x.is_contiguous() and cu.dtype == torch.int32
    W = cu.numel() - 1
    out = torch.empty((W, x.shape[1]), dtype=out_dtype, device=x.device)
    check(_lib.load().vf_segment_mean(x.data_ptr(), cu.data_ptr(), out.data_ptr(), W, x.shape[1], _dt(out_dtype),
                                      _stream()), "vf_segment_mean")
    return out


def segment_mean16(x16: torch.Tensor, cu: torch.Tensor, in_scale: float = 1.0, split: bool = False) -> torch.Tensor:
    """Per-window mean of a 16-bit stream [n_tok, d] (rows may be strided), values times in_scale (vf_segment_mean16):
    fp32 [W, d], or with split=True the means as two 16-bit halves [W, 2 d] = [hi | lo] (lo = rn16(mean - hi)) in the type
    of x16 -- a 16-bit GEMM operand that carries the fp32 mean to ~2^-17."""
    _dev(x16, cu)
    assert _is16(x16.dtype) and x16.dim() == 2 and x16.stride(1) == 1 and cu.dtype == torch.int32
    W, d = cu.numel() - 1, x16.shape[1]
    out = torch.empty((W, 2 * d), dtype=x16.dtype, device=x16.device) if split else torch.empty((W, d), dtype=torch.float32,
                                                                                               device=x16.device)
    check(_lib.load().vf_segment_mean16(x16.data_ptr(), x16.stride(0), _dt(x16.dtype), cu.data_ptr(), float(in_scale),
                                        None if split else out.data_ptr(), out.data_ptr() if split else None, W, d,
                                        _stream()), "vf_segment_mean16")
    return out


def segment_max(x: torch.Tensor, cu: torch.Tensor) -> torch.Tensor:
    _dev(x, cu)
    assert x.dtype == torch.float32 and x.is_contiguous() and cu.dtype == torch.int32
    W = cu.numel() - 1
    out = torch.empty((W, x.shape[1]), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_segment_max(x.data_ptr(), cu.data_ptr(), out.data_ptr(), W, x.shape[1], _stream()),
          "vf_segment_max")
    return out


def segment_linear(x: torch.Tensor, cu: torch.Tensor, pad: torch.Tensor, lin_w: torch.Tensor, lin_b: torch.Tensor | None,
                   out_dtype=None) -> torch.Tensor:
    """seq2reg "linear" pooling: out[w] = sum_p lin_w[p] * x[row(w, p)] over the valid positions p, + lin_b."""
    _dev(x, cu, pad, lin_w, lin_b)
    out_dtype = _CDT if out_dtype is None else out_dtype
    pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad
    assert x.dtype == torch.float32 and x.is_contiguous() and cu.dtype == torch.int32 and pad.dtype == torch.uint8
    W, L = pad.shape
    assert lin_w.dtype == torch.float32 and lin_w.numel() == L and lin_w.is_contiguous() and cu.numel() == W + 1
    out = torch.empty((W, x.shape[1]), dtype=out_dtype, device=x.device)
    check(_lib.load().vf_segment_linear(x.data_ptr(), cu.data_ptr(), pad.data_ptr(), lin_w.data_ptr(), _ptr(lin_b),
                                        out.data_ptr(), W, L, x.shape[1], _dt(out_dtype), _stream()), "vf_segment_linear")
    return out


def affine_rows(src: torch.Tensor, idx: torch.Tensor, scale: torch.Tensor | None = None,
                shift: torch.Tensor | None = None) -> torch.Tensor:
    """out[i] = src[idx[i]] * scale[i] + shift[i] (fp32 rows; scale / shift optional per-row scalars)."""
    _dev(src, idx, scale, shift)
    assert src.dtype == torch.float32 and src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    for t in (scale, shift):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == idx.numel())
    out = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    check(_lib.load().vf_affine_rows_f32(src.data_ptr(), idx.data_ptr(), _ptr(scale), _ptr(shift), out.data_ptr(),
                                         idx.numel(), src.shape[1], _stream()), "vf_affine_rows_f32")
    return out


def add_rows(a: torch.Tensor, b: torch.Tensor, idx_a: torch.Tensor | None = None, idx_b: torch.Tensor | None = None) -> torch.Tensor:
    """out[i] = a[idx_a[i]] + b[idx_b[i]] (identity where an index is None), fp32 rows."""
    _dev(a, b, idx_a, idx_b)
    for t in (a, b):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2
    for t in (idx_a, idx_b):
        assert t is None or (t.dtype == torch.int64 and t.is_contiguous())
    n = idx_a.numel() if idx_a is not None else (idx_b.numel() if idx_b is not None else a.shape[0])
    assert a.shape[1] == b.shape[1] and (idx_a is not None or a.shape[0] == n) and (idx_b is not None or b.shape[0] == n)
    out = torch.empty((n, a.shape[1]), dtype=torch.float32, device=a.device)
    check(_lib.load().vf_add_rows_f32(a.data_ptr(), _ptr(idx_a), b.data_ptr(), _ptr(idx_b), out.data_ptr(), n, a.shape[1],
                                      _stream()), "vf_add_rows_f32")
    return out


def gather_rows_f32(a: torch.Tensor, b: torch.Tensor | None, idx: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    """out[i] = a[idx[i]] if idx[i] >= 0 else b[-idx[i]-1]."""
    _dev(a, b, idx)
    assert a.dtype == torch.float32 and a.is_contiguous() and idx.dtype == torch.int64
    assert b is None or (b.dtype == torch.float32 and b.is_contiguous() and b.shape[1] == a.shape[1])
    out = torch.empty((idx.numel(), a.shape[1]), dtype=out_dtype, device=a.device)
    check(_lib.load().vf_gather_rows_f32(a.data_ptr(), _ptr(b), idx.data_ptr(), out.data_ptr(), idx.numel(), a.shape[1],
                                         _dt(out_dtype), _stream()), "vf_gather_rows_f32")
    return out


def gather_rows_bf16(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _dev(src, idx)
    assert _is16(src.dtype) and src.stride(1) == 1 and idx.dtype == torch.int64
    out = torch.empty((idx.numel(), src.shape[1]), dtype=src.dtype, device=src.device)

    def launch():
        check(_lib.load().vf_gather_rows_bf16(src.data_ptr(), src.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0),
                                              idx.numel(), src.shape[1], _stream()), "vf_gather_rows_bf16")
    if TIMER is not None:
        TIMER.time("layernorm", 0.0, float(idx.numel()) * (src.shape[1] * 4 + 8), launch, f"gather_rows16 D={src.shape[1]}", _SCOPE)
    else:
        launch()
    return out


def rowdot_softplus(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, softplus: bool = True) -> torch.Tensor:
    _dev(x, w, b)
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == torch.float32 and w.numel() == x.shape[1]
    out = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_rowdot_softplus(x.data_ptr(), w.data_ptr(), _ptr(b), out.data_ptr(), x.shape[0], x.shape[1],
                                         int(softplus), _stream()), "vf_rowdot_softplus")
    return out


def last_kernel(which: str = "gemm") -> str:
    """Name of the kernel this thread's most recent GEMM ("gemm") or attention ("attn") entry dispatched to (vf_last_kernel:
    a diagnostic for sweeps over shapes, scripts/s2r_dims_sweep.py)."""
    name = _lib.load().vf_last_kernel(0 if which == "gemm" else 1)
    return name.decode() if name else ""


def cast16(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """fp32 -> 16-bit operand type (default: the current compute dtype), round to nearest even."""
    _dev(x)
    dtype = _CDT if dtype is None else dtype
    assert x.dtype == torch.float32 and x.is_contiguous() and _is16(dtype)
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    lib = _lib.load()
    fn = lib.vf_cast_f32_f16 if dtype == torch.float16 else lib.vf_cast_f32_bf16
    check(fn(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "vf_cast_f32_16")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    return cast16(x, torch.bfloat16)
