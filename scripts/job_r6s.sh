#!/bin/bash
# round 6: soak with the CRE side stream as the default -- 300 timed steps (drift, allocations, peak memory), 1 024 genes through the
# product flow from genome files, the dh = 96 attention test and the whole attention file once more
O=gpurun_out/r6s
mkdir -p $O
timeout 900 python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extra-rates --no-pipelined --no-kernel-timing > $O/soak.json 2> $O/soak.err; echo "soak rc $?"
python - <<'PY'
import json
for l in open("gpurun_out/r6s/soak.json"):
    if l.startswith("{"):
        d=json.loads(l); ms=d["ms_of_each_timed_step_rank0"]
        n=len(ms); a=sum(ms[:20])/20; b=sum(ms[-20:])/20
        print(f"300 steps: value {d['value']} genes/s, {d['ms_per_step']} ms/step; first 20 steps {a:.2f} ms, last 20 steps {b:.2f} ms ({(b/a-1)*100:+.2f} %), min {min(ms):.2f} max {max(ms):.2f}; device allocations inside the timed region {d['device_allocations_inside_timed_region']}; peak HBM {d['peak_hbm_allocated_gb']} GB")
PY
timeout 1200 python scripts/vcf2exp_e2e.py --genes 1024 --workers 3 --batch 32 --cores 4 --passes 1 2>&1 | grep "^pass\|memory\|peak" ; echo "e2e rc $?"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "dh96 or one_block" 2>&1 | tail -2
