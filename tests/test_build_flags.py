"""The shipped library's device code has no packed-fp32 VALU instructions (csrc/build.py NO_PACKED_FP32: on gfx950 an
op_sel:[0,1,..] form of them goes wrong in lanes 48..63 beside another kernel's MFMAs -- profiles/r06_d_pk_hazard_probe.log)."""
import os
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_library_has_no_packed_fp32_instructions(tmp_path):
    from variantformer_amd.csrc import build as B
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not present")
    lib = B.build_lib()
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)     # unbundles beside the copy
    images = [f for f in os.listdir(tmp_path) if f.endswith(B.ARCH)]
    assert len(images) >= 3, images                                   # vf_gemm, vf_attn, vf_misc
    n_mfma = 0
    for f in images:
        dis = subprocess.run([OBJDUMP, "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        n_mfma += dis.count("v_mfma_f32_")
        for ins in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            assert ins not in dis, f"{ins} in the device code of {f}"
        # the only instruction left that carries an op_sel operand selection is v_fma_mix_f32 (fp32 += float(half) * x), which the
        # probe shows unaffected, with the half in either half of its register; anything else must be probed before it ships
        others = {line.split()[0] for line in dis.splitlines() if "op_sel" in line and line.split()} - {"v_fma_mix_f32"}
        assert not others, f"instructions with op_sel other than v_fma_mix_f32 in {f}: {sorted(others)}"
    assert n_mfma > 1000                                               # (the disassembly really is the kernels)
