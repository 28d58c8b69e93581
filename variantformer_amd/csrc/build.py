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


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


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
               "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + defines + EXTRA_FLAGS.get(src, []) + \
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
