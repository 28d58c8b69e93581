#!/bin/bash
# residency probes of gemm4_kernel (tuning library): default, LDS request forced to one block per CU, grid of one block per CU
export VF_TUNING_LIB=1 VARIANTS=22,40
for f in "" "VF_G4_LDS=102400" "VF_G4_BPC=1" "VF_G4_STAGGER=40" "VF_G4_STAGGER=120"; do
  echo "== $f"
  env $f python scripts/gemm4_probe.py time "$1" 2>&1 | grep -v amdgpu.ids
done
