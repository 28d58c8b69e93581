"""Interleaved A/B of two builds of the library on the workload's LayerNorm-folded GEMM shapes (child processes alternate,
same box): python scripts/ab_gemm_lib.py libvf_hip_prev.so [genes]   -- TFLOP/s per shape and build, best of 5."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    other = sys.argv[1]
    g = sys.argv[2] if len(sys.argv) > 2 else "8"
    for rep in range(2):
        for lib in (other, "libvf_hip.so"):
            r = subprocess.run([sys.executable, __file__, "--child", lib, g], capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-500:], flush=True)
    sys.exit(0)
import torch
from variantformer_amd import ops, _lib
lib, g = sys.argv[2], int(sys.argv[3])
_lib.load(os.path.join(os.path.dirname(_lib.LIB_PATH), lib))
Mg, Ms = 10854 * g, 96183 * g
shapes = [("gene Wqkv cons", Mg, 4608, 1536, "c"), ("gene GeGLU cons", Mg, 2048, 1536, "g"), ("gene Wq cons", Mg, 1536, 1536, "c"),
          ("gene out_proj r16", Mg, 1536, 1536, "p"), ("s2r Wqkv cons", Ms, 1536, 512, "c"), ("s2r GeGLU cons", Ms, 2048, 512, "g"),
          ("s2r out_proj r16", Ms, 512, 512, "p")]
out = []
for name, M, N, K, kind in shapes:
    a = (torch.rand((M, K), device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand((N,), device="cuda")
    if kind == "p":
        res = ops.ln_stream(torch.rand((M, N), device="cuda"))
        f = lambda: ops.gemm_ln_producer(a, w, b, res, need_x=False)
    else:
        s = ops.ln_stream(a.float())
        if kind == "g":
            w, b = ops.pack_geglu_rows(w, b)
        cs = w.float().sum(1).contiguous()
        epi = ops.EPI_GEGLU_BF16 if kind == "g" else ops.EPI_BF16
        f = lambda: ops.gemm_ln_consumer(s, w, b, cs, epi)
    for _ in range(2):
        f()
    best = 1e9
    for _ in range(5):
        st, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); f(); e.record(); torch.cuda.synchronize()
        best = min(best, st.elapsed_time(e))
    out.append("%s %.0f" % (name.replace(" ", "_"), 2.0 * M * N * K / best / 1e9))
    del a, w, b
    torch.cuda.empty_cache()
print("%-20s " % lib + "  ".join(out))
