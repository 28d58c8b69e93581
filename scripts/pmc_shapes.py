"""HBM traffic PER LAUNCH GEOMETRY of the bench step (round-5 verdict, Next 1a: the aggregate over all gemm8* launches says
1.85x the algorithmic bytes, but not which shapes carry it -- the persistent kernels all launch 256 blocks, so neither the
kernel name nor the grid tells the shapes apart).  The launches of a step are deterministic and every ops.gemm* / ops.attn*
call is exactly one kernel dispatch, so the i-th GEMM (attention) dispatch of a rocprofv3 counter pass IS the i-th "gemm"
("attn") record of ops.KernelTimer(detail=True).order of the same program.

  run   (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace, one pass each, the program itself after `--`):
        python3 scripts/pmc_shapes.py run <genes_per_step> <order.json>
        one untimed + one recorded step of the headline workload with the timer on from the first launch; writes the order.
  join  python scripts/pmc_shapes.py join <fetch_dir> <write_dir> <order.json> <out.json>
        FETCH_SIZE x 2 (gfx950: 64 B tallied per 128-B request, MI355X_MICROARCH.md section HBM) and WRITE_SIZE, KiB -> bytes,
        per geometry: launches, HBM read / written per launch, algorithmic bytes per launch (the timer's model), their ratio.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GEMM_KERNELS = ("gemm8x_kernel", "gemm8y_kernel", "gemm8_kernel", "gemm_mfma_kernel", "gemm_generic_kernel")
ATTN_KERNELS = ("attn_", "softmax_counted_kernel")


def run(genes: int, order_path: str):
    import torch
    import bench
    from variantformer_amd import ops
    from variantformer_amd.utils.synthetic import TISSUES_54, make_batch
    dev = torch.device("cuda:0")
    ops.TIMER = ops.KernelTimer(detail=True)            # from the FIRST launch: table builds and weight packing dispatch too
    model, hp, kw = bench.build_model(dev)
    batch = make_batch(20251205, [1024] * genes, [200] * genes, [TISSUES_54] * genes, 200)
    with torch.no_grad():
        pb = model.prepare_batch(batch)
        model.forward_prepared(pb)
        torch.cuda.synchronize()
        first = len(ops.TIMER.order)
        model.forward_prepared(pb)
        torch.cuda.synchronize()
    order = ops.TIMER.order
    ops.TIMER = None
    json.dump({"genes": genes, "steady_from": first, "order": order, "source_sha": bench.source_sha()}, open(order_path, "w"))
    print("launches recorded:", len(order), "steady step from", first)


def dispatches(d: str, counter: str):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = int(r["Dispatch_Id"])
            e = rows.setdefault(k, [r["Kernel_Name"], 0.0])
            e[1] += float(r["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(rows.items())]


def join(fetch_dir, write_dir, order_path, out_path):
    o = json.load(open(order_path))
    order, first = o["order"], o["steady_from"]
    res = {"_source_sha": o["source_sha"], "_genes_per_step": o["genes"],
           "_method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/pmc_shapes.py run; dispatch i of a kernel "
                      "family = launch i of ops.KernelTimer.order; FETCH_SIZE x2, KiB -> bytes; steady-state step only"}
    for kind, names in (("gemm", GEMM_KERNELS), ("attn", ATTN_KERNELS)):
        recs = [(i, r) for i, r in enumerate(order) if r[0] == kind]
        per = {}
        for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
            disp = [x for x in dispatches(d, counter) if any(n in x[1] for n in names)]
            if len(disp) != len(recs):
                raise SystemExit(f"{kind}: {len(disp)} {counter} dispatches vs {len(recs)} timer records -- the 1:1 assumption broke")
            for (i, r), (_, kname, val) in zip(recs, disp):
                if i < first:
                    continue
                short = kname.replace("void (anonymous namespace)::", "").split("(")[0]
                e = per.setdefault((r[1], short), {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "alg_bytes": 0.0, "family": set()})
                if counter == "FETCH_SIZE":
                    e["launches"] += 1
                    e["fetch_kib"] += val
                    e["alg_bytes"] += r[4]
                    e["family"].add(r[2])
                else:
                    e["write_kib"] += val
        rows = []
        for (geom, kname), e in per.items():
            n = max(e["launches"], 1)
            rd, wr, alg = 2.0 * e["fetch_kib"] * 1024 / n, e["write_kib"] * 1024 / n, e["alg_bytes"] / n
            rows.append({"geometry": geom, "kernel": kname, "family": sorted(e["family"]), "launches_per_step": e["launches"],
                         "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "algorithmic_bytes_per_launch": alg,
                         "traffic_over_algorithmic": (rd + wr) / alg if alg else None,
                         "excess_bytes_per_step": (rd + wr - alg) * e["launches"]})
        rows.sort(key=lambda r: -r["excess_bytes_per_step"])
        res[kind] = rows
        tot_alg = sum(r["algorithmic_bytes_per_launch"] * r["launches_per_step"] for r in rows)
        tot = sum((r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]) * r["launches_per_step"] for r in rows)
        res[f"_{kind}_total"] = {"hbm_bytes_per_step": tot, "algorithmic_bytes_per_step": tot_alg, "ratio": tot / tot_alg if tot_alg else None}
    json.dump(res, open(out_path, "w"), indent=1)
    for kind in ("gemm", "attn"):
        print(f"== {kind}: total {res['_' + kind + '_total']}")
        for r in res[kind][:24]:
            print(f"{r['geometry']:62s} {r['kernel'][:34]:34s} n={r['launches_per_step']:3d} read {r['hbm_read_bytes_per_launch'] / 1e9:7.3f} GB "
                  f"write {r['hbm_write_bytes_per_launch'] / 1e9:7.3f} GB alg {r['algorithmic_bytes_per_launch'] / 1e9:7.3f} GB "
                  f"x{r['traffic_over_algorithmic']:.2f}  excess/step {r['excess_bytes_per_step'] / 1e9:7.2f} GB")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3])
    else:
        join(*sys.argv[2:6])
