#!/bin/bash
# round 6, final set on the final build: profile set (GPU suite, bench, rocprofv3 kernel stats, PMC passes incl. per launch
# geometry, fp16, configs[2], shape breakdown), the seq2reg geometry sweep, the product flow from genome files on 2 / 1 host cores
bash scripts/run_profile_set.sh r6z
O=gpurun_out/r6z
timeout 1500 python scripts/s2r_dims_sweep.py $O/s2r_dims.json > $O/s2r_dims.log 2>&1; echo "dims sweep rc $?"; tail -16 $O/s2r_dims.log
for c in 2 1; do
  timeout 900 python scripts/vcf2exp_e2e.py --genes 256 --workers 1 --batch 32 --cores $c --passes 2 2>&1 | grep "^pass\|cores" >> $O/host_capacity.log
done
timeout 900 python scripts/vcf2exp_e2e.py --genes 256 --workers 3 --batch 32 --cores 4 --passes 2 --overlap 2>&1 | grep "^pass" >> $O/host_capacity.log
timeout 900 python scripts/vcf2exp_e2e.py --genes 256 --workers 3 --batch 32 --cores 4 --passes 2 --overlap --no-dedupe 2>&1 | grep "^pass" >> $O/host_capacity.log
cat $O/host_capacity.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
