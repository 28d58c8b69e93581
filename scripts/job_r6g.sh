#!/bin/bash
# round 6, job g: concurrency probe 4 over the library builds
mkdir -p gpurun_out/r6g
for lib in "" variantformer_amd/csrc/probe_libs/libvf_nopk_gemm.so variantformer_amd/csrc/probe_libs/libvf_wait0.so variantformer_amd/csrc/probe_libs/libvf_nopk_attn.so; do
  timeout 300 python scripts/probes/concurrency_probe4.py $lib >> gpurun_out/r6g/concurrency_probe4.log 2>&1
  echo "---- rc $?" >> gpurun_out/r6g/concurrency_probe4.log
done
tail -60 gpurun_out/r6g/concurrency_probe4.log
