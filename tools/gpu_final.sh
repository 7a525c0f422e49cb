# Round-end measurement set (GPU box): bench lines, secondary measurements, rocprofv3 stats + counters.
# usage: bash tools/gpu_final.sh <tag>   -> gpurun_out/final_<tag>/
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --no-pipeline --no-cpu-baseline > $OUT/bench_n1_strict_stream_order.json 2> $OUT/bench_strict.err
python bench.py --mode incremental --no-cpu-baseline > $OUT/bench_incremental_n1.json 2> $OUT/bench_incremental_n1.err
python bench.py --batch 4194304 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_n1_batch4M.json 2> $OUT/bench_4m.err
python tools/gpu_measure.py > $OUT/secondary.json 2> $OUT/secondary.err
python tools/spx_selfplay.py --games 4096 --target 8192 > $OUT/selfplay_4096.json 2> $OUT/selfplay.err
python tools/spx_selfplay.py --games 16384 --target 32768 > $OUT/selfplay_16384.json 2>> $OUT/selfplay.err
python tools/spx_selfplay.py --games 1024 --target 2048 > $OUT/selfplay_1024.json 2>> $OUT/selfplay.err
python tools/gpu_movegen_rate.py > $OUT/movegen_rate.json 2>> $OUT/selfplay.err
python tools/spx_selfplay.py --games 4096 --target 8192 --host-movegen > $OUT/selfplay_4096_host_movegen.json 2>> $OUT/selfplay.err
python tools/gpu_latency.py > $OUT/latency.txt 2>&1
bash tools/gpu_profile.sh $TAG > /dev/null 2>&1
cp $REPO/gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt
# (the TA/TD/TCP counter groups are NOT collected here: on 2026-09-28 rocprofv3 aborted inside hipMemcpy with them and
#  then hung in its signal handler until the timeout - every rocprofv3 call in tools/ now runs under `timeout 300`)
bash tools/gpu_stats.sh default_$TAG > $OUT/rocprofv3_kernel_stats_default_cmd.txt 2>&1
bash tools/gpu_stats.sh inc_$TAG --mode incremental > $OUT/rocprofv3_incremental_kernel_stats.txt 2>&1
ls -la $OUT
