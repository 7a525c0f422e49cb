# Round-end measurement set (GPU box): bench lines, secondary measurements, rocprofv3 stats + counters.
# usage: bash tools/gpu_final.sh <tag>   -> gpurun_out/final_<tag>/   (copy what is quoted into profiles/)
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
# counters FIRST: bench.py quotes the VALU / L2 counter blocks from profiles/<tag>_pmc_*.json, so the files of THIS build
# are put in place (on the box) before the bench lines are taken
bash tools/gpu_profile.sh $TAG > /dev/null 2>&1
cp $REPO/gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary_full_refresh.txt
cp $REPO/gpurun_out/prof_$TAG/pmc.json $OUT/pmc_full_refresh.json
bash tools/gpu_pmc_inc.sh $TAG > /dev/null 2>&1
cp $REPO/gpurun_out/pmc_inc_$TAG/summary.txt $OUT/rocprofv3_summary_incremental.txt
cp $REPO/gpurun_out/pmc_inc_$TAG/pmc.json $OUT/pmc_incremental.json
rm -rf $REPO/gpurun_out/prof_$TAG $REPO/gpurun_out/pmc_inc_$TAG
[ -s $OUT/pmc_full_refresh.json ] && cp $OUT/pmc_full_refresh.json $REPO/profiles/${TAG}_pmc_full_refresh.json
[ -s $OUT/pmc_incremental.json ] && cp $OUT/pmc_incremental.json $REPO/profiles/${TAG}_pmc_incremental.json
( time python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err ) 2> $OUT/bench_n1_wall_time.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_args.json 2> $OUT/bench_driver.err
python bench.py --no-pipeline --no-cpu-baseline > $OUT/bench_n1_strict_stream_order.json 2> $OUT/bench_strict.err
python bench.py --device-positions --no-cpu-baseline --no-wide > $OUT/bench_n1_device_positions.json 2> $OUT/bench_devpos.err
python bench.py --mode incremental --no-cpu-baseline > $OUT/bench_incremental_n1.json 2> $OUT/bench_incremental_n1.err
python bench.py --mode incremental --no-pipeline --no-cpu-baseline > $OUT/bench_incremental_n1_strict_stream_order.json 2>> $OUT/bench_incremental_n1.err
python bench.py --mode incremental --batch 262144 --steps 100 --no-cpu-baseline > $OUT/bench_incremental_n1_262144.json 2>> $OUT/bench_incremental_n1.err
python bench.py --batch 4194304 --steps 20 --warmup 3 --no-cpu-baseline --no-wide > $OUT/bench_n1_batch4M.json 2> $OUT/bench_4m.err
python tools/gpu_measure.py > $OUT/secondary.json 2> $OUT/secondary.err
python tools/spx_selfplay.py --games 4096 --target 8192 --dfrc > $OUT/selfplay_4096.json 2> $OUT/selfplay.err
python tools/spx_selfplay.py --games 4096 --target 65536 --dfrc > $OUT/selfplay_4096_long.json 2>> $OUT/selfplay.err
python tools/spx_selfplay.py --games 16384 --target 65536 --dfrc > $OUT/selfplay_16384.json 2>> $OUT/selfplay.err
python tools/spx_selfplay.py --games 1024 --target 8192 --dfrc > $OUT/selfplay_1024.json 2>> $OUT/selfplay.err
# round 5: the live fixed-node search in the driver (64 expansions per search: whole games; 1 000: games cut
# at 16 plies so that the run stays bounded), the replay's segment length
python tools/spx_selfplay.py --games 4096 --target 4096 --dfrc --temperature 0 --search-nodes 64 > $OUT/selfplay_search_64_nodes.json 2>> $OUT/selfplay.err
python tools/spx_selfplay.py --games 4096 --target 4096 --dfrc --temperature 0 --search-nodes 1000 --max-plies 16 > $OUT/selfplay_search_1000_nodes.json 2>> $OUT/selfplay.err
# (datagen's 24 000 nodes: 4 minutes for 1 024 games of 8 plies - run once by hand, profiles/r05_selfplay_search_24000_nodes_1024_seats_8_plies.json)
python tools/gpu_replay_segment_ab.py > $OUT/replay_segment_ab.txt 2>&1
./stormphrax_amd/spx_raweval --preset tame --walk 7 6000 rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 > $OUT/raweval_walk.txt 2>&1
bash tools/gpu_selfplay_busy.sh 4096 32768 > $OUT/selfplay_gpu_busy.txt 2>&1
python tools/gpu_movegen_rate.py > $OUT/movegen_rate.json 2>> $OUT/selfplay.err
python tools/gpu_latency.py > $OUT/latency.txt 2>&1
python tools/gpu_replay_rate.py > $OUT/config3_replay.json 2> $OUT/replay.err
# (the TA/TD/TCP counter groups are NOT collected here: on 2026-09-28 rocprofv3 aborted inside hipMemcpy with them and
#  then hung in its signal handler until the timeout - every rocprofv3 call in tools/ runs under `timeout 300`)
# differential runs against the COMPILED REFERENCE on the final binary (its probe travels in oracle/_ref/)
python tools/gpu_ref_differential.py --positions 10000000 --pack-positions 1000000 > $OUT/reference_differential.json 2> $OUT/differential.err
python tools/gpu_ref_trace_differential.py --roots 64 > $OUT/reference_trace_differential.json 2>> $OUT/differential.err
# counters of the self-play's eval-only update kernel (what crosses the fabric per child now)
( cd /tmp && export TMPDIR=/tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/sp_pmc_$(echo $grp | cut -d' ' -f1) -o c -- python $REPO/tools/spx_selfplay.py --games 4096 --target 8192 --dfrc > /dev/null 2>&1
  done )
python3 - <<PY > $OUT/pmc_selfplay_4096_seats.txt 2>&1
import glob, sqlite3
print("rocprofv3 --pmc <group> --kernel-trace -- python tools/spx_selfplay.py --games 4096 --target 8192 --dfrc  (mean per dispatch)")
for f in sorted(glob.glob("$OUT/sp_pmc_*/*.db")):
    c = sqlite3.connect(f).cursor()
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if "rocclr" in r[0]: continue
        print("   %-44s %-18s mean/dispatch %16.1f  dispatches %d" % (r[0].replace("spx::", "")[:44], r[1], r[2], r[3]))
PY
rm -rf $OUT/sp_pmc_*
# round 4: the N > 1 line's new legs (two ranks sharing this GPU over gloo: control flow, not a scaling number), BASELINE configs[4],
# the column-sliced pipeline's kernels, the reference engine on the GPU evaluator
SPX_BENCH_SHARE_GPU=1 SPX_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-wide > $OUT/bench_n2_two_ranks_sharing_one_gpu.json 2> $OUT/bench_n2.err
python tests/_config5_worker.py > $OUT/config5.log 2>&1 && cp $REPO/gpurun_out/config5_hbm_filling.json $OUT/bench_n1_config5_hbm_filling.json
bash tools/gpu_kstats.sh sliced_$TAG --no-pipeline > $OUT/kstats_sliced_pipeline_stream_ordered.txt 2>&1
SPX_OPTIONS=ftx=0 bash tools/gpu_kstats.sh onekernel_$TAG --no-pipeline > $OUT/kstats_one_kernel_path_stream_ordered.txt 2>&1
bash tools/gpu_pmc_ftx.sh $TAG > /dev/null 2>&1; cp $REPO/gpurun_out/pmc_ftx_$TAG.txt $OUT/pmc_sliced_pipeline.txt
bash tools/gpu_timeline.sh > $OUT/timeline_pipelined_steps.txt 2>&1
bash tools/gpu_ftx_crossover.sh > $OUT/sliced_pipeline_crossover.txt 2>&1
SPX_OPTIONS=ftx=0 python tools/gpu_ref_differential.py --positions 4000000 --pack-positions 100000 > $OUT/reference_differential_one_kernel_path.json 2>> $OUT/differential.err
( for mode in cpu gpu; do $REPO/oracle/_ref/sp_ref_gpu_tame bench 6 $mode | grep -v "^info\|^fen:\|^bestmove\|^$"; done
  $REPO/oracle/_ref/sp_ref_gpu_tame bench 4 both | grep -v "^info\|^fen:\|^bestmove\|^$"
  $REPO/oracle/_ref/sp_ref_gpu_tame game 3000 5 ) > $OUT/reference_engine_on_gpu_evaluator.txt 2>&1
bash tools/gpu_stats.sh default_$TAG > $OUT/rocprofv3_kernel_stats_default_cmd.txt 2>&1
bash tools/gpu_stats.sh headline_$TAG --no-secondary --no-wide > $OUT/rocprofv3_kernel_stats_headline_only.txt 2>&1
bash tools/gpu_stats.sh inc_$TAG --mode incremental > $OUT/rocprofv3_incremental_kernel_stats.txt 2>&1
rm -rf $REPO/gpurun_out/stats_default_$TAG $REPO/gpurun_out/stats_headline_$TAG $REPO/gpurun_out/stats_inc_$TAG
ls -la $OUT
