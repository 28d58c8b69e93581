#!/bin/bash
# round 6, job j: the library without packed-fp32 instructions -- probes 4/5 on it and on the pk_* builds, the GPU suite with the
# default switches and with the CRE side stream on, bench both ways
mkdir -p gpurun_out/r6j
L=variantformer_amd/csrc/probe_libs
O=gpurun_out/r6j
timeout 300 python scripts/probes/concurrency_probe4.py > $O/concurrency_probe4_product.log 2>&1; echo "probe4 product rc $?"
timeout 300 python scripts/probes/concurrency_probe5.py > $O/concurrency_probe5_product.log 2>&1; echo "probe5 product rc $?"
timeout 300 python scripts/probes/with_lib.py $L/libvf_pk_all.so scripts/probes/concurrency_probe5.py > $O/concurrency_probe5_pk_all.log 2>&1; echo "probe5 pk_all rc $?"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest default rc $?"; tail -3 $O/pytest_gpu.log
timeout 1500 python scripts/probes/with_switches.py overlap_cre_stream=1 -m pytest tests -m gpu -q > $O/pytest_gpu_overlap_on.log 2>&1; echo "pytest overlap rc $?"; tail -8 $O/pytest_gpu_overlap_on.log
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 900 python scripts/probes/with_switches.py overlap_cre_stream=1 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-rates > $O/bench_overlap_on.json 2> $O/bench_overlap_on.err; echo "bench overlap rc $?"
timeout 900 python scripts/probes/overlap_diag.py 64 > $O/overlap_diag.log 2>&1; echo "overlap diag rc $?"; tail -5 $O/overlap_diag.log
grep -c "bit-identical" $O/concurrency_probe4_product.log $O/concurrency_probe5_product.log $O/concurrency_probe5_pk_all.log
grep -h "WRONG" $O/concurrency_probe4_product.log $O/concurrency_probe5_product.log $O/concurrency_probe5_pk_all.log | cut -c1-220
for f in $O/bench.json $O/bench_overlap_on.json; do python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cre_side_stream_experiment"), d.get("batch_of_8"), d.get("batch_of_1"), d.get("pipelined_product_flow", d.get("product_flow")))
PY
done
