#!/bin/bash
# genes/s of the product flow (genome files -> expression) against the host cores one rank may use: 1 / 2 / 4 / 14.
# usage (GPU box, repo root): bash scripts/host_capacity.sh <out.log> [genes] [batch]
OUT=${1:-gpurun_out/host_capacity.log}; G=${2:-128}; B=${3:-32}
: > $OUT
for c in 14 4 2 1; do   # C host cores for the whole rank: C - 1 loader workers + the main thread (1 worker at C = 1)
  w=$c; [ $c -gt 1 ] && w=$((c - 1))
  python scripts/vcf2exp_e2e.py --genes $G --workers $w --batch $B --cores $c --passes 2 2>&1 | grep "^pass" >> $OUT
done
python scripts/vcf2exp_e2e.py --genes $G --workers 13 --batch $B --cores 14 --passes 2 --overlap 2>&1 | grep "^pass" >> $OUT
python scripts/vcf2exp_e2e.py --genes $G --workers 13 --batch $B --cores 14 --passes 2 --overlap --no-dedupe 2>&1 | grep "^pass" >> $OUT
cat $OUT
