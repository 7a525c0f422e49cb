# The bench lines, kernel statistics and counters of the full-refresh path alone (after a change that touches nothing else):
# bash tools/gpu_final_refresh.sh <tag>  -> gpurun_out/final_<tag>/ (same names as tools/gpu_final.sh; install with install_final_profiles.py)
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
bash tools/gpu_profile.sh $TAG > /dev/null 2>&1
cp $REPO/gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary_full_refresh.txt
cp $REPO/gpurun_out/prof_$TAG/pmc.json $OUT/pmc_full_refresh.json
rm -rf $REPO/gpurun_out/prof_$TAG
[ -s $OUT/pmc_full_refresh.json ] && cp $OUT/pmc_full_refresh.json $REPO/profiles/${TAG}_pmc_full_refresh.json
( time python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err ) 2> $OUT/bench_n1_wall_time.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_args.json 2> $OUT/bench_driver.err
python bench.py --no-pipeline --no-cpu-baseline > $OUT/bench_n1_strict_stream_order.json 2> $OUT/bench_strict.err
python bench.py --device-positions --no-cpu-baseline --no-wide > $OUT/bench_n1_device_positions.json 2> $OUT/bench_devpos.err
python bench.py --batch 4194304 --steps 20 --warmup 3 --no-cpu-baseline --no-wide > $OUT/bench_n1_batch4M.json 2> $OUT/bench_4m.err
bash tools/gpu_kstats.sh sliced_$TAG --no-pipeline > $OUT/kstats_sliced_pipeline_stream_ordered.txt 2>&1
bash tools/gpu_pmc_ftx.sh $TAG > /dev/null 2>&1; cp $REPO/gpurun_out/pmc_ftx_$TAG.txt $OUT/pmc_sliced_pipeline.txt
bash tools/gpu_timeline.sh > $OUT/timeline_pipelined_steps.txt 2>&1
bash tools/gpu_ftx_crossover.sh > $OUT/sliced_pipeline_crossover.txt 2>&1
bash tools/gpu_stats.sh default_$TAG > $OUT/rocprofv3_kernel_stats_default_cmd.txt 2>&1
bash tools/gpu_stats.sh headline_$TAG --no-secondary --no-wide > $OUT/rocprofv3_kernel_stats_headline_only.txt 2>&1
rm -rf $REPO/gpurun_out/stats_default_$TAG $REPO/gpurun_out/stats_headline_$TAG
ls -la $OUT | head -30
