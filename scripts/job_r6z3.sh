#!/bin/bash
# round 6, final set part 3: dh = 96 / 128 windows as one 128-query block (sweep of d = 768 / 1024 against the default), the
# attention tests, then the rocprofv3 passes on this build
O=gpurun_out/r6z
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attn or attention" > $O/pytest_attention.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_attention.log
timeout 1200 python scripts/s2r_dims_sweep.py $O/s2r_dims_wide_heads.json --widths=512,768,1024 > $O/s2r_dims_wide_heads.log 2>&1; echo "sweep rc $?"
grep "dh= 96\|dh=128" $O/s2r_dims_wide_heads.log | awk 'NR%6==1' | cut -c1-200
tail -14 $O/s2r_dims_wide_heads.log | cut -c1-200
bash scripts/run_profile_set.sh r6z --prof-only > $O/prof_only.log 2>&1; tail -5 $O/prof_only.log | cut -c1-200
