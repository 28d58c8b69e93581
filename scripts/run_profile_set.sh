# usage (on the GPU box, from the repo root): bash scripts/run_profile_set.sh <tag> [--skip-tests | --pmc-only | --prof-only]
# (--prof-only: only the rocprofv3 passes -- kernel stats + PMC -- e.g. after a change of the profiled command line)
# Writes gpurun_out/<tag>/: pytest log, bench line, rocprofv3 kernel stats, HBM-traffic PMC passes, SQ PMC pass.
# Every rocprofv3 line has the program itself after `--` (no env / bash -c hop) and never mixes --pmc with trace domains
# other than --kernel-trace.
TAG=${1:-x}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "--skip-tests" ] && [ "$2" != "--pmc-only" ] && [ "$2" != "--prof-only" ]; then
  python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log; grep -a "\[headline\|\[trained-like" $O/pytest_gpu.log
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$2" != "--pmc-only" ] && [ "$2" != "--prof-only" ]; then
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
fi
if [ "$2" != "--pmc-only" ]; then
# kernel stats: W + K + K identical steps in the ONE-stream order (with the CRE side stream a kernel's traced duration includes the
# time it shares the CUs with the other stream's kernels; `roofline` in every bench line comes from the one-stream replay), and
# the same command in the default two-stream order beside it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extra-rates --no-pipelined --single-stream > $O/bench_prof.json 2> $O/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 bench.py --no-cpu-baseline --no-extra-rates --no-pipelined > $O/bench_prof_two_streams.json 2> $O/prof2.err
cp $(find $O/prof2 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_two_streams.csv 2>/dev/null; rm -rf $O/prof2
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-rates --no-pipelined > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-rates --no-pipelined > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-rates --no-pipelined > $O/pmc_sq.json 2> $O/pmc_sq.err
python scripts/pmc_summary.py $O/pmc_hbm_traffic.json ${TAG}_pmc > /dev/null
# per launch geometry (round 6): the same two counters on scripts/pmc_shapes.py, joined with the timer's launch order
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_shp_fetch -- python3 scripts/pmc_shapes.py run 32 $O/pmc_shapes_order.json > $O/pmc_shapes_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_shp_write -- python3 scripts/pmc_shapes.py run 32 $O/pmc_shapes_order.json > $O/pmc_shapes_write.log 2>&1
python scripts/pmc_shapes.py join gpurun_out/${TAG}_shp_fetch gpurun_out/${TAG}_shp_write $O/pmc_shapes_order.json $O/pmc_hbm_traffic_by_shape.json > $O/pmc_hbm_traffic_by_shape.txt 2>&1; head -40 $O/pmc_hbm_traffic_by_shape.txt
rm -rf gpurun_out/${TAG}_shp_fetch gpurun_out/${TAG}_shp_write $O/pmc_shapes_order.json
python scripts/pmc_sq_summary.py gpurun_out/${TAG}_pmc_sq $O/pmc_sq.json > $O/pmc_sq_summary.txt; tail -60 $O/pmc_sq_summary.txt
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_sq
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
du -sh $O
if [ "$2" != "--pmc-only" ] && [ "$2" != "--prof-only" ]; then
python bench.py --dtype fp16 --no-cpu-baseline --no-pipelined > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/bench_cfg3_1gpu.json 2>> $O/bench.err
python scripts/shape_breakdown.py 32 > $O/shape_breakdown.txt 2>&1
fi
