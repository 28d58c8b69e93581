"""Build libvf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m variantformer_amd.csrc.build [--force]

The shared object is written next to the sources (in-tree, git-ignored) so that it travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["vf_gemm.hip", "vf_attn.hip", "vf_misc.hip", "vf_bpe.cpp", "vf_vcf.cpp", "vf_host.cpp"]
HEADERS = ["vf_common.h", os.path.join("..", "..", "include", "vf_hip.h")] + \
    [os.path.join("tuning", f) for f in ("gemm_persist.inc", "gemm4.inc", "gemm8p.inc", "gemm_xs.inc", "gemm8y.inc", "attn_x32pp.inc")]
LIB = os.path.join(HERE, "libvf_hip.so")
ARCH = "gfx950"
# vf_attn: scores are never NaN by construction (finite inputs, -inf only as a mask), so fmaxf needs no
# canonicalising v_max in front of every max (68 extra VALU instructions per key tile otherwise).
EXTRA_FLAGS = {"vf_attn.hip": ["-fno-honor-nans"]}
# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any device code.  Measured on MI355X in
# round 6 (scripts/probes/pk_hazard_probe.hip, profiles/r06_d_pk_hazard_probe.log): a packed-fp32 instruction whose op_sel
# takes the HIGH dword of src1 for the low result (op_sel:[0,1,..] -- what hipcc's SLP vectoriser emits for `vec * scalar`,
# e.g. the LayerNorm-consumer epilogue acc * rstd + t) computes with a wrong src1 in lanes 48..63 while a wave of ANOTHER
# kernel on the same SIMD issues MFMAs (up to 0.6 % of the results beside dense v_mfma_f32_16x16x32_bf16; never alone, never
# beside anything else, never for the other op_sel / op_sel_hi forms or for scalar v_fma_f32).  In one stream kernels do not
# overlap, so the default path never met it; two streams did (the CRE side-stream experiment's 7e-4 run-to-run differences,
# LayerNorm-consumer GEMMs beside the cross attention: scripts/probes/concurrency_probe4.py).  The compiler chooses the op_sel
# form, so the instructions are switched off as a whole: results are bit-identical (a v_pk_fma_f32 is two v_fma_f32), the GEMM
# epilogues cost +0.9 % GEMM time (profiles/r06_d_nopk_cost.txt).  tests/test_build_flags.py keeps the library free of them.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + [os.path.basename(__file__)])   # flags live here


TUNING_LIB = os.path.join(HERE, "libvf_hip_tuning.so")


def build_lib(force: bool = False, verbose: bool = False, tuning: bool = False) -> str:
    """tuning=True builds libvf_hip_tuning.so with -DVF_TUNING: the diagnostic kernel variants and the tile sweep of
    scripts/ (results of the diagnostic variants are meaningless).  The product library never contains them."""
    if tuning:
        return _build(TUNING_LIB, ["-DVF_TUNING"], verbose, "tuning_")
    if not force and not _stale():
        return LIB
    return _build(LIB, [], verbose, "")


def _build(LIB: str, defines: list, verbose: bool, obj_prefix: str) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvf_hip.so")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, obj_prefix + os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
               "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + (NO_PACKED_FP32 if src.endswith(".hip") else []) + defines + \
              EXTRA_FLAGS.get(src, []) + \
              ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out:
            print(out)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + ["-lz"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv, tuning="--tuning" in sys.argv))
