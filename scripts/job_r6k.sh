#!/bin/bash
# round 6, job k: CRE side stream as the default -- the two tests that pin it, probe 5 on the product / pk_all builds, bench
O=gpurun_out/r6k
mkdir -p $O
L=variantformer_amd/csrc/probe_libs
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ln_heal_gpu.py tests/test_concurrency_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_subset.log
timeout 300 python scripts/probes/concurrency_probe5.py > $O/concurrency_probe5_product.log 2>&1; echo "probe5 product rc $?"
timeout 300 python scripts/probes/with_lib.py $L/libvf_pk_all.so scripts/probes/concurrency_probe5.py > $O/concurrency_probe5_pk_all.log 2>&1; echo "probe5 pk_all rc $?"
grep -c "bit-identical" $O/concurrency_probe5_product.log $O/concurrency_probe5_pk_all.log
grep -h "WRONG" $O/concurrency_probe5_product.log $O/concurrency_probe5_pk_all.log | cut -c1-220
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for l in open("gpurun_out/r6k/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["warmup"], d["roofline"]["frac"], d.get("value_single_stream"), d.get("batch_of_8"), d.get("batch_of_1"), d.get("vcf2exp_pipelined"), d.get("ln_fold_off"), d.get("trained_like"))
PY
