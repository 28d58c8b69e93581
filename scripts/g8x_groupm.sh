#!/bin/bash
# tile-walk sweep of the persistent 256x256 kernel (tuning library): GROUP_M m-panels x all n, m fastest; 1 = n fastest
export VF_TUNING_LIB=1 VARIANTS=22
for gm in 8 1 2 3 4 6 16 32; do
  echo "== VF_G8X_GROUP_M=$gm"
  VF_G8X_GROUP_M=$gm python scripts/gemm4_probe.py time 2>&1 | grep -v amdgpu.ids | grep -v "^shape"
done
