"""Round 6: alternative builds of libvf_hip.so for scripts/probes/concurrency_probe4.py / concurrency_probe5.py (written to
variantformer_amd/csrc/probe_libs/, git-ignored, travel to the GPU box; run a script on one with scripts/probes/with_lib.py).
The product build has NO packed-fp32 VALU instructions (csrc/build.py NO_PACKED_FP32); each variant re-admits them in one object:
    pk_gemm : vf_gemm.hip WITH v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (the build of rounds 1-5: reproduces the hazard)
    pk_attn : vf_attn.hip with them
    pk_all  : vf_gemm, vf_attn and vf_misc with them
(Probe 4's log also lists a `wait0` build: vf_gemm.hip of that day with an s_waitcnt vmcnt(0) behind the LayerNorm consumer's
epilogue loads -- it changed nothing and the hook is gone.)
"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from variantformer_amd.csrc import build as B

B.build_lib()
OUT = os.path.join(B.HERE, "probe_libs")
os.makedirs(OUT, exist_ok=True)
base = ["/opt/rocm/bin/hipcc", f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def obj(src, tag, extra):
    o = os.path.join(OUT, f"{tag}_{os.path.splitext(src)[0]}.o")
    subprocess.run(base + extra + B.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(B.HERE, src), "-o", o], check=True)
    return o


def link(name, repl):
    objs = [repl.get(s, os.path.join(B.HERE, os.path.splitext(s)[0] + ".o")) for s in B.SOURCES]
    lib = os.path.join(OUT, f"libvf_{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-lz"], check=True)
    print(lib)


g, a, m = (obj(s, "pk", []) for s in ("vf_gemm.hip", "vf_attn.hip", "vf_misc.hip"))
link("pk_gemm", {"vf_gemm.hip": g})
link("pk_attn", {"vf_attn.hip": a})
link("pk_all", {"vf_gemm.hip": g, "vf_attn.hip": a, "vf_misc.hip": m})
