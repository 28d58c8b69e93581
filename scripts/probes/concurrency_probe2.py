"""Round 6, follow-up of concurrency_probe.py: WHICH output goes wrong when two kernels run on two streams at once, where and by
how much.  Pairs that mismatched: gene->CRE cross attention beside the CRE-stream consumer GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from variantformer_amd import ops
from scripts.probes.concurrency_probe import cases, refs, names   # builds the cases and their single-stream references

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def describe(tag, out, ref):
    if torch.equal(out, ref):
        return None
    d = (out.float() - ref.float()).abs()
    bad = d > 0
    rows = torch.nonzero(bad.any(dim=1)).flatten()
    cols = torch.nonzero(bad.any(dim=0)).flatten()
    return (f"{tag}: {int(bad.sum())} of {bad.numel()} elements differ, max |diff| {float(d.max()):.3e} (ref max {float(ref.float().abs().max()):.3e}); "
            f"rows {int(rows.min())}..{int(rows.max())} ({rows.numel()} rows; 256-row tiles {sorted(set((rows // 256).tolist()))[:12]}), "
            f"cols {int(cols.min())}..{int(cols.max())} ({cols.numel()} cols; 256-col tiles {sorted(set((cols // 256).tolist()))[:12]})")


pairs = [("gene->CRE cross attention", "CRE Wqkv consumer (1024 x G rows)"), ("gene->CRE cross attention", "CRE low-rank logits (N = 320, fp32 out)"),
         ("gene out_proj producer r16 (gemm8x)", "CRE Wqkv consumer (1024 x G rows)"), ("gene->CRE cross attention", "CRE GeGLU consumer"),
         ("CRE Wqkv consumer (1024 x G rows)", "gene->CRE cross attention")]
with torch.no_grad():
    for a, b in pairs:
        print(f"== main: {a}  ||  side: {b}   [{ops.last_kernel('gemm')} / {ops.last_kernel('attn')}]", flush=True)
        for rep in range(4):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ob = [cases[b]() for _ in range(3)]
            oa = cases[a]()
            main.wait_stream(side)
            torch.cuda.synchronize()
            msgs = [describe("  main output", oa, refs[a])] + [describe(f"  side output {i}", x, refs[b]) for i, x in enumerate(ob)]
            msgs = [m for m in msgs if m]
            print(f" round {rep}: " + ("all bit-identical" if not msgs else ""), flush=True)
            for m in msgs:
                print(m, flush=True)
    # the same launches back to back on ONE stream (control)
    for a, b in pairs[:2]:
        oa = cases[a](); ob = cases[b](); torch.cuda.synchronize()
        print(f"control, one stream: {a} {'ok' if torch.equal(oa, refs[a]) else 'DIFFERS'}; {b} {'ok' if torch.equal(ob, refs[b]) else 'DIFFERS'}")
