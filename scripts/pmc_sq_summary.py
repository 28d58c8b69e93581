"""Summarise a rocprofv3 --pmc pass over SQ counters into per-kernel totals and derived ratios.

    python scripts/pmc_sq_summary.py <dir with *counter_collection.csv> <out.json>

Units (MI355X_MICROARCH.md "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES
count quad-cycles summed over waves (resp. over SQs); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per 16x16x32 bf16 MFMA).
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES x 4): the share of SIMD-cycles with the matrix pipe busy.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out_path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(gemm\w+|attn_\w+|layernorm_kernel|seq2reg_\w+)(<[^>]*(?:<[^>]*>[^>]*)*>)?", name)
        if not m:
            continue
        key = (m.group(1) + (m.group(2) or ""))[:90]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[key].add((f, r["Dispatch_Id"]))
# kernel durations of the same pass (--kernel-trace): MFMA-busy cycles per SIMD per microsecond, i.e. the matrix pipe's
# busy time expressed as a clock rate (2400 = busy every cycle at the 2.4 GHz maximum clock; the chip holds ~1.9-2.1 GHz
# under MFMA load, MI355X_MICROARCH.md "DVFS give-back")
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm\w+|attn_\w+|layernorm_kernel|seq2reg_\w+)(<[^>]*(?:<[^>]*>[^>]*)*>)?", r["Kernel_Name"])
        if m:
            dur[(m.group(1) + (m.group(2) or ""))[:90]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
res = {}
for k, c in sorted(acc.items()):
    n = len(launches[k])
    e = {"launches": n}
    e.update({cn: v / n for cn, v in c.items()})
    busy = c.get("SQ_BUSY_CYCLES", 0.0)
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    if dur.get(k):
        e["avg_duration_us"] = dur[k] / n
        e["mfma_busy_MHz_per_simd"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * dur[k])
        e["mfma_busy_frac_of_2400MHz"] = e["mfma_busy_MHz_per_simd"] / 2400.0
    if busy:
        e["mfma_busy_over_sq_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (busy * 16.0)
    if wave:
        e["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / wave
        e["wait_inst_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wave
        e["active_inst_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    res[k] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
res["_source_sha"] = bench.source_sha()
res["_method"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- python3 bench.py --steps 1 "
                  "--warmup 1 --no-cpu-baseline --no-kernel-timing; per-launch averages per kernel instantiation")
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                      if "frac" in kk or kk in ("launches", "avg_duration_us", "mfma_busy_MHz_per_simd")} for k, v in res.items() if isinstance(v, dict)}, indent=1))
