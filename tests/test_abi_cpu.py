"""CPU-side checks of the drop-in boundary: libvf_hip.so builds for gfx950, loads, and exports exactly
the entry points include/vf_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import REPO


def _declared():
    with open(os.path.join(REPO, "include", "vf_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from variantformer_amd.csrc.build import build_lib
    from variantformer_amd import _lib
    path = build_lib()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in vf_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert _lib.load().vf_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Error behaviour of the boundary: invalid arguments are rejected before any launch."""
    from variantformer_amd import _lib
    lib = _lib.load()
    assert lib.vf_gemm_bf16(0, 64, 0, 0, 0, 0, 0, 64, 4, 8, 64, 0, 0) == 1          # null pointers
    assert b"null" in lib.vf_last_error()
    assert lib.vf_attn_varlen_fwd(16, 16, 16, 16, 8, 8, 8, 8, 16, 16, 1, 1, 1, 1, 40, 0, 1.0, 0) == 1   # dh=40
    assert b"head_dim" in lib.vf_last_error()
    assert lib.vf_layernorm(16, 16, 16, 16, 1, 6, 1e-5, 1, 0, 0) == 1                 # D % 4 != 0
    with pytest.raises(_lib.VFError):
        _lib.check(1, "demo")


def test_ops_refuse_cpu_tensors():
    import torch
    from variantformer_amd import ops
    from variantformer_amd._lib import VFError
    with pytest.raises(VFError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))


def test_vf_narrow_ids_clamps_narrows_and_reports_ranges():
    """vf_narrow_ids (ABI 11, host only): int64 ids -> int32, clamped to [-1, INT32_MAX] like the embedding kernels' own clamp
    would treat them, with the range flags prepare_batch keys its window de-duplication on; strided source rows."""
    import numpy as np
    from variantformer_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    src = rng.integers(0, 500, (7, 3, 40)).astype(np.int64)          # [rows, strands, L]: rows are 3 * 40 elements apart
    view = src[:, 1, :]
    dst = np.full((7, 40), -7, dtype=np.int32)
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 7, 40) == 0
    assert np.array_equal(dst, view.astype(np.int32))
    view[2, 5], view[4, 0] = -(2 ** 40), -1
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 7, 40) == 1
    assert dst[2, 5] == -1 and dst[4, 0] == -1
    view[2, 5], view[4, 0], view[6, 39] = 3, 4, 2 ** 31 + 9
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 7, 40) == 2
    assert dst[6, 39] == 2 ** 31 - 1
    view[6, 39], view[0, 0] = 2 ** 30, -5
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 7, 40) == 3 and dst[6, 39] == 2 ** 30
    view[6, 39], view[0, 0] = 2 ** 30 - 1, 0
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 7, 40) == 0
    assert lib.vf_narrow_ids(None, 0, dst.ctypes.data, 7, 40) == -1
    assert lib.vf_narrow_ids(view.ctypes.data, view.strides[0] // 8, dst.ctypes.data, 0, 40) == 0
