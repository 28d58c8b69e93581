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
