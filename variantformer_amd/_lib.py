"""ctypes binding of libvf_hip.so (C ABI declared in include/vf_hip.h).

The product path has NO fallback: if the shared library is missing, cannot be loaded, or lacks a
declared symbol, importing the ops raises.  Signatures are plain pointers and sizes; torch only
supplies device memory (tensor.data_ptr()) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvf_hip.so")

# enums of include/vf_hip.h
VF_F32, VF_BF16, VF_F16 = 0, 1, 2
EPI_BF16, EPI_F32, EPI_RES_F32, EPI_GEGLU_BF16, EPI_GELU_F32, EPI_GELU_BF16 = 0, 1, 2, 3, 4, 5
ABI_VERSION = 12

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype is int unless noted); mirrors include/vf_hip.h one to one
SIGNATURES = {
    "vf_version": [],
    "vf_last_error": [],
    "vf_last_kernel": [_i],
    "vf_gemm_bf16": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _p],
    "vf_gemm_bf16_ex": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "vf_gemm_f16": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _p],
    "vf_gemm_f16_ex": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "vf_gemm_ln_bf16": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _p, _p, _p, _l, _p, _p],
    "vf_gemm_ln_t16": [_p, _l, _p, _p, _p, _l, _f, _p, _l, _i, _i, _i, _i, _p, _l, _p, _f, _p, _l, _f, _p],
    "vf_gemm_ln": [_p, _l, _p, _p, _p, _l, _i, _p, _l, _i, _i, _i, _i, _i, _p, _p, _p, _l, _p, _f, _f, _p],
    "vf_ln_finalize": [_p, _l, _i, _i, _f, _p, _p],
    "vf_ln_finalize2": [_p, _l, _i, _i, _f, _f, _f, _f, _p, _p, _p],
    "vf_row_stats_cast2": [_p, _l, _i, _f, _p, _i, _f, _f, _f, _p, _p, _p],
    "vf_row_stats_cast": [_p, _l, _i, _f, _p, _i, _p, _p],
    "vf_pack_geglu_rows": [_p, _p, _p, _p, _i, _i, _p],
    "vf_attn_varlen_fwd": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p],
    "vf_attn_varlen_fwd_qstart": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p],
    "vf_attn_varlen_fwd_f16": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p],
    "vf_attn_varlen_fwd_qstart_f16": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p],
    "vf_attn_varlen_fwd_v2": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _i, _i, _p],
    "vf_attn_rows_supported": [_i, _i, _i, _i, _i, _i, _i],
    "vf_attn_counted_keys": [_p, _l, _p, _l, _p, _p, _i, _i, _i, _i, _i, _p, _l, _i, _p],
    "vf_softmax_counted": [_p, _l, _p, _p, _i, _i, _i, _i, _i, _p, _l, _i, _p],
    "vf_attn_varlen_fwd_rows": [_p, _p, _p, _p, _l, _l, _l, _l, _p, _p, _i, _i, _i, _i, _i, _p, _f, _i, _i, _p, _p, _p],
    "vf_layernorm": [_p, _p, _p, _p, _l, _i, _f, _i, _i, _p],
    "vf_embed_stream": [_p, _p, _p, _p, _p, _p, _p, _i, _f, _p, _f, _p, _f, _f, _f, _p, _i, _i, _i, _i, _p],
    "vf_embed_pack": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vf_mask_to_cu_seqlens": [_p, _p, _i, _i, _p],
    "vf_segment_mean": [_p, _p, _p, _i, _i, _i, _p],
    "vf_segment_mean16": [_p, _l, _i, _p, _f, _p, _p, _i, _i, _p],
    "vf_token_keys": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vf_gather_rows_f32": [_p, _p, _p, _p, _l, _i, _i, _p],
    "vf_gather_rows_bf16": [_p, _l, _p, _p, _l, _l, _i, _p],
    "vf_rowdot_softplus": [_p, _p, _p, _p, _l, _i, _i, _p],
    "vf_cast_f32_bf16": [_p, _p, _l, _p],
    "vf_cast_f32_f16": [_p, _p, _l, _p],
    "vf_segment_max": [_p, _p, _p, _i, _i, _p],
    "vf_segment_linear": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vf_affine_rows_f32": [_p, _p, _p, _p, _p, _l, _i, _p],
    "vf_add_rows_f32": [_p, _p, _p, _p, _p, _l, _i, _p],
    "vf_bpe_create": [_p, _i, _p, _i],
    "vf_bpe_destroy": [_p],
    "vf_bpe_encode": [_p, C.c_char_p, _l, _p, _p, _l],
    "vf_bpe_encode_prefix": [_p, C.c_char_p, _l, _l, _p, _p, _l],
    "vf_vcf_open": [C.c_char_p, C.c_char_p],
    "vf_vcf_close": [_p],
    "vf_vcf_num_records": [_p, C.c_char_p],
    "vf_vcf_consensus": [_p, C.c_char_p, _l, C.c_char_p, _l, _i, _i, _p, _l, _p],
    "vf_build_windows": [_p, _p, C.c_char_p, _l, C.c_char_p, _l, _l, _p, _p, _i, _i, _i, _i, _l, _p, _p, _p],
    "vf_narrow_ids": [_p, _l, _p, _l, _l],
}
_RESTYPES = {"vf_last_error": C.c_char_p, "vf_last_kernel": C.c_char_p, "vf_bpe_create": C.c_void_p, "vf_bpe_destroy": None, "vf_bpe_encode": C.c_int64, "vf_bpe_encode_prefix": C.c_int64,
             "vf_vcf_open": C.c_void_p, "vf_vcf_close": None, "vf_vcf_num_records": C.c_int64,
             "vf_vcf_consensus": C.c_int64, "vf_build_windows": C.c_int64}

_lib = None


class VFError(RuntimeError):
    pass


def load(path: str | None = None):
    """Load libvf_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # torch ships its own libamdhip64; import it FIRST so that libvf_hip.so binds to the HIP runtime torch
    # already loaded.  Loaded the other way round the process ends up with two HIP runtimes and the second
    # one to initialise reports "no ROCm-capable device".
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise VFError(
            f"{path} not found: the HIP extension has not been built "
            "(run `python -m variantformer_amd.csrc.build` or __graft_entry__.build()). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VFError(f"libvf_hip.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = lib.vf_version()
    if v != ABI_VERSION:
        raise VFError(f"libvf_hip.so ABI version {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().vf_last_error()
        raise VFError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
