#!/bin/bash
# round 6, job h: which kernels are at risk beside another kind of kernel; cost of a library without packed-fp32 VALU
# instructions; the CRE side-stream experiment on that library
# NOTE: libvf_nopk_* / libvf_wait0 were built by the first version of scripts/probes/build_probe_libs.py (the round-5 flags plus
# -target-feature -packed-fp32-ops on ONE object; wait0 = a temporary s_waitcnt vmcnt(0) hook).  The product build now carries the
# flag itself and the script builds the inverse variants (libvf_pk_*): the same comparison with the roles swapped.
mkdir -p gpurun_out/r6h
L=variantformer_amd/csrc/probe_libs
timeout 400 python scripts/probes/concurrency_probe5.py > gpurun_out/r6h/concurrency_probe5.log 2>&1
echo "---- rc $?" >> gpurun_out/r6h/concurrency_probe5.log
timeout 400 python scripts/probes/with_lib.py $L/libvf_nopk_all.so scripts/probes/concurrency_probe5.py > gpurun_out/r6h/concurrency_probe5_nopk_all.log 2>&1
echo "---- rc $?" >> gpurun_out/r6h/concurrency_probe5_nopk_all.log
for lib in nopk_gemm nopk_all; do
  timeout 900 python scripts/probes/with_lib.py $L/libvf_$lib.so bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined > gpurun_out/r6h/bench_$lib.json 2> gpurun_out/r6h/bench_$lib.err
  echo "bench $lib rc $?"
done
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined > gpurun_out/r6h/bench_default.json 2> gpurun_out/r6h/bench_default.err
echo "bench default rc $?"
timeout 900 python scripts/probes/with_lib.py $L/libvf_nopk_gemm.so scripts/probes/overlap_diag.py 64 > gpurun_out/r6h/overlap_diag_nopk_gemm.log 2>&1
echo "overlap diag rc $?"
tail -30 gpurun_out/r6h/concurrency_probe5.log
tail -5 gpurun_out/r6h/overlap_diag_nopk_gemm.log
for f in gpurun_out/r6h/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cre_side_stream_experiment"), d.get("ln_fold_off"))
PY
done
