#!/bin/bash
# round 6, job g: concurrency probe 4 over the library builds
# NOTE: libvf_nopk_* / libvf_wait0 were built by the first version of scripts/probes/build_probe_libs.py (the round-5 flags plus
# -target-feature -packed-fp32-ops on ONE object; wait0 = a temporary s_waitcnt vmcnt(0) hook).  The product build now carries the
# flag itself and the script builds the inverse variants (libvf_pk_*): the same comparison with the roles swapped.
mkdir -p gpurun_out/r6g
for lib in "" variantformer_amd/csrc/probe_libs/libvf_nopk_gemm.so variantformer_amd/csrc/probe_libs/libvf_wait0.so variantformer_amd/csrc/probe_libs/libvf_nopk_attn.so; do
  timeout 300 python scripts/probes/concurrency_probe4.py $lib >> gpurun_out/r6g/concurrency_probe4.log 2>&1
  echo "---- rc $?" >> gpurun_out/r6g/concurrency_probe4.log
done
tail -60 gpurun_out/r6g/concurrency_probe4.log
