"""Performance over the seq2reg (tokenizer) geometries the real checkpoint might have.

The tokenizer's width / heads / depth / positional kind / pooling / use_context live in `pretrained_tokenizers_checkpoint.pth`
(reference processors/model_manager.py:44-51,77; configs/vf_model.yaml:38-41), which is unavailable offline: bench.py and every
perf number assume d = 512 / 8 heads / 6 layers / sinusoidal / mean.  This sweep runs ONE 8-gene headline step (N = 1024 cCRE
windows, C = 200 gene chunks, T = 54 tissues; the seq2gene stack is the shipped 25-layer / 1536-wide one throughout) for
    d in {256, 512, 768, 1024} x head_dim in {32, 64, 96, 128} (where heads = d / head_dim is an integer)
      x layers in {4, 6, 12} x {sinusoidal, alibi} x {mean, max}     (+ use_context=True CRE tokenizers at the default width)
and records per geometry: genes/s of the step, the seq2reg families' time and achieved TFLOP/s (GEMM) / TB/s (attention)
against their own FLOP / byte model, which GEMM / attention kernels were dispatched (vf_last_kernel: no gemm_generic_kernel,
which attention kernel, whether the first-layer lookup gathered inside the attention or by a row gather first), and the
first-layer lookup table's bytes.  Output: one JSON (profiles/r06_s2r_dims.json) + a table on stdout.

usage: python scripts/s2r_dims_sweep.py [out.json] [--quick] [--widths=512,768]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from variantformer_amd import ops  # noqa: E402
from variantformer_amd.seq2gene.model_combined_modulator import Seq2GenePredictorCombinedModulator  # noqa: E402
from variantformer_amd.seq2reg.model import Seq2RegPredictor  # noqa: E402
from variantformer_amd.utils.synthetic import TISSUES_54, make_batch  # noqa: E402

out_path = next((a for a in sys.argv[1:] if not a.startswith("--")), "gpurun_out/s2r_dims.json")
quick = "--quick" in sys.argv
G = 8
dev = torch.device("cuda:0")


def init_(module, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if p.dim() >= 2 and "embedding" not in name and "registry" not in name:
                p.normal_(0.0, 0.4 / float(p.shape[-1]) ** 0.5, generator=g)
            elif p.dim() >= 2:
                p.normal_(0.0, 0.29, generator=g)
            elif leaf == "weight":
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.01, generator=g)


def tokenizer(d, heads, layers, pos, pool, use_context=False):
    hp = dict(bench.SEQ2REG_HP, embedding_dim=d, num_heads=heads, num_layers=layers, positional_encoding=pos, seq_pool=pool,
              use_context=use_context)
    with torch.device(dev):
        t = Seq2RegPredictor(**hp)
    if pos == "sinusoidal":
        t.position_encoding = t.position_encoding.cpu()
    init_(t, 7)
    return t.eval()


class Spy:
    """Records (family, geometry) -> kernel name for every GEMM / attention launch (ops.KernelTimer hook + vf_last_kernel)."""

    def __init__(self):
        self.seen = {}
        self.timer = ops.KernelTimer(detail=True)

    def time(self, name, flops, nbytes, launch, geometry="", family=""):
        out = self.timer.time(name, flops, nbytes, launch, geometry, family)
        if name in ("gemm", "attn"):
            self.seen.setdefault((name, family, geometry), ops.last_kernel(name))
        return out

    def summary(self):
        return self.timer.summary()


batch = make_batch(20251205, [1024] * G, [200] * G, [TISSUES_54] * G, 200)
rows = []
widths = [512] if quick else [512, 256, 768, 1024]
for a in sys.argv[1:]:
    if a.startswith("--widths="):                    # e.g. --widths=512,768 (512 must be there: the default geometry is the yardstick)
        widths = [int(x) for x in a.split("=", 1)[1].split(",")]
for d in widths:
    kw = dict(bench.SEQ2GENE_KW, token_dim=d, gene_emb_dim=d)
    with torch.device(dev):
        model = Seq2GenePredictorCombinedModulator(cre_tokenizer=None, gene_tokenizer=None, **kw)
    init_(model, 1234)
    model.eval()
    configs = []
    for dh in (32, 64, 96, 128):
        if d % dh:
            continue
        for layers in ((6,) if quick else (4, 6, 12)):
            for pos in ("sinusoidal", "alibi"):
                for pool in ("mean", "max"):
                    configs.append((d // dh, layers, pos, pool, False))
    if d == 512:
        configs += [(8, 6, "sinusoidal", "mean", True), (8, 6, "alibi", "mean", True)]
    for heads, layers, pos, pool, ctx in configs:
        cre_tok = tokenizer(d, heads, layers, pos, pool, ctx)
        gene_tok = tokenizer(d, heads, layers, pos, pool, False)          # (the reference's gene branch cannot use a context tokenizer)
        model.cre_tokenizer, model.gene_tokenizer = cre_tok, gene_tok
        with torch.no_grad():
            pb = model.prepare_batch(batch)
            for _ in range(2):
                model.forward_prepared(pb)[0].cpu()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.forward_prepared(pb)[0].cpu()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            spy = Spy()
            ops.TIMER = spy
            model.forward_prepared(pb)[0].cpu()
            summ = spy.summary()
            ops.TIMER = None
        fam = {k: v for k, v in summ.items() if ":" in k and k.split(":", 1)[1].startswith("seq2reg")}
        gemm = fam.get("gemm:seq2reg", {"total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
        attn = fam.get("attn:seq2reg_self", {"total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
        other = sum(v["total_ms"] for k, v in fam.items() if k not in ("gemm:seq2reg", "attn:seq2reg_self"))
        kernels = sorted({f"{n}:{k}" for (n, f, g), k in spy.seen.items() if f.startswith("seq2reg")})
        attn_geoms = sorted({g for (n, f, g), k in spy.seen.items() if n == "attn" and f.startswith("seq2reg")})
        g_inten = gemm["flops"] / max(gemm["bytes"], 1.0)                        # flop per algorithmic byte of the family's GEMMs
        g_roof = min(2500.0, g_inten * 8.0)                                      # TFLOP/s: min(dense bf16 MFMA peak, intensity x 8 TB/s)
        g_ach = gemm["flops"] / max(gemm["total_ms"], 1e-9) / 1e9
        rec = {"seq2reg_gemm_intensity_flop_per_byte": round(g_inten, 1), "seq2reg_gemm_roofline_TFLOPs": round(g_roof, 1),
               "seq2reg_gemm_frac_of_roofline": round(g_ach / g_roof, 4),
               "seq2reg_attn_frac_of_hbm_roofline": round(attn["bytes"] / max(attn["total_ms"], 1e-9) / 1e9 / 8.0, 4)}
        rec.update({"d": d, "heads": heads, "head_dim": d // heads, "layers": layers, "positional": pos, "pool": pool, "use_context": ctx,
               "genes_per_s": round(G / dt, 3), "ms_per_step": round(dt * 1e3, 2),
               "seq2reg_gemm_ms": round(gemm["total_ms"], 2), "seq2reg_gemm_TFLOPs": round(gemm["flops"] / max(gemm["total_ms"], 1e-9) / 1e9, 1),
               "seq2reg_attn_ms": round(attn["total_ms"], 2), "seq2reg_attn_TBps": round(attn["bytes"] / max(attn["total_ms"], 1e-9) / 1e9, 3),
               "seq2reg_attn_TFLOPs": round(attn["flops"] / max(attn["total_ms"], 1e-9) / 1e9, 1),
               "seq2reg_other_ms": round(other, 2), "kernels": kernels, "attention_launch_forms": attn_geoms,
               "layer0_table_bytes": int(cre_tok._layer0_qkv_table_bytes()),
               "layer0_lookup_used": bool(getattr(cre_tok, "_qkv_tabs", None))})
        rows.append(rec)
        print(f"d={d:4d} h={heads:2d} dh={d // heads:3d} L={layers:2d} {pos[:4]} {pool:4s} ctx={int(ctx)}  {rec['genes_per_s']:7.2f} genes/s  "
              f"gemm {rec['seq2reg_gemm_ms']:7.2f} ms {rec['seq2reg_gemm_TFLOPs']:6.0f} TF/s  attn {rec['seq2reg_attn_ms']:6.2f} ms "
              f"{rec['seq2reg_attn_TBps']:5.2f} TB/s  other {rec['seq2reg_other_ms']:5.2f} ms  {' '.join(k.split(':', 1)[1] for k in kernels)}",
              flush=True)
        model.cre_tokenizer = model.gene_tokenizer = None
        del cre_tok, gene_tok, pb
        torch.cuda.empty_cache()
    del model
    torch.cuda.empty_cache()

base = next(r for r in rows if (r["d"], r["heads"], r["layers"], r["positional"], r["pool"], r["use_context"]) == (512, 8, 6, "sinusoidal", "mean", False))
for r in rows:
    # the geometry's fraction of ITS OWN roofline (GEMM: min(MFMA peak, intensity x HBM); attention: HBM) over the default geometry's
    r["gemm_rate_vs_default"] = round(r["seq2reg_gemm_frac_of_roofline"] / base["seq2reg_gemm_frac_of_roofline"], 3)
    r["attn_rate_vs_default"] = round(r["seq2reg_attn_frac_of_hbm_roofline"] / base["seq2reg_attn_frac_of_hbm_roofline"], 3)
worst = sorted(rows, key=lambda r: min(r["gemm_rate_vs_default"], r["attn_rate_vs_default"]))[:12]
json.dump({"genes_per_step": G, "default": base, "rows": rows, "source_sha": bench.source_sha(),
           "note": "rates vs the default geometry's: each seq2reg family's fraction of its own roofline (GEMM: min(2.5 PFLOP/s, "
                   "intensity x 8 TB/s) on the algorithmic bytes; attention: algorithmic bytes / 8 TB/s) over the default's"},
          open(out_path, "w"), indent=1)
print("\nlowest fractions of the own roofline relative to the default geometry's:")
for r in worst:
    print(f"  d={r['d']} dh={r['head_dim']} L={r['layers']} {r['positional']} {r['pool']} ctx={int(r['use_context'])}: "
          f"gemm x{r['gemm_rate_vs_default']:.2f} attn x{r['attn_rate_vs_default']:.2f}  {r['genes_per_s']} genes/s")
