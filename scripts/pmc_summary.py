"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE exact; both in KiB."""
import csv, glob, json, sys, collections
out_path, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launches_f": 0, "launches_w": 0})
for f in glob.glob("gpurun_out/%s_fetch/**/*counter_collection.csv" % tag, recursive=True) + \
         glob.glob("gpurun_out/%s_write/**/*counter_collection.csv" % tag, recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = ("gemm8_kernel" if ("gemm8_kernel" in name or "gemm8x_kernel" in name) else "gemm_mfma_kernel" if "gemm_mfma_kernel" in name else
               ("attn_kernels" if "attn_" in name else ("layernorm_kernel" if "layernorm" in name else None)))
        if key is None:
            continue
        keys = [key]
        if key == "attn_kernels":          # also one entry per attention kernel instantiation (name up to the argument list)
            keys.append(name.replace("void (anonymous namespace)::", "").split("(")[0])
        d = (f, r["Dispatch_Id"])
        for key in keys:
            if r["Counter_Name"] == "FETCH_SIZE":
                acc[key]["fetch_kib"] += float(r["Counter_Value"])
                if d not in seen: acc[key]["launches_f"] += 1
            elif r["Counter_Name"] == "WRITE_SIZE":
                acc[key]["write_kib"] += float(r["Counter_Value"])
                if d not in seen: acc[key]["launches_w"] += 1
        seen.add(d)
res = {}
for k, v in acc.items():
    lf, lw = max(v["launches_f"], 1), max(v["launches_w"], 1)
    res[k] = {"launches": lf, "hbm_read_bytes_per_launch": 2.0 * v["fetch_kib"] * 1024 / lf,
              "hbm_write_bytes_per_launch": v["write_kib"] * 1024 / lw}
    res[k]["hbm_bytes_per_launch"] = res[k]["hbm_read_bytes_per_launch"] + res[k]["hbm_write_bytes_per_launch"]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res["_source_sha"] = bench.source_sha()
res["_method"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; FETCH_SIZE x2 (gfx950), KiB -> bytes"
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
