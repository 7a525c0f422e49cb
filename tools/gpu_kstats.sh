# per-kernel durations of a bench run: bash tools/gpu_kstats.sh <tag> [bench args]   -> gpurun_out/kstats_<tag>.txt
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o t -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-settle --no-wide --no-secondary "$@" > $OUT/stats.log 2>&1
python3 - <<PY > $REPO/gpurun_out/kstats_$TAG.txt 2>&1
import glob, sqlite3
for f in glob.glob("$OUT/stats/*.db"):
    c = sqlite3.connect(f).cursor()
    print("== rocprofv3 --kernel-trace --stats of: bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-settle --no-wide --no-secondary $*")
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("   %-60s calls %5d total %10.1f us avg %9.2f us  %5.1f%%" % (r[0][:60], r[1], r[2] / 1e3 if r[3] > 1e4 else r[2], r[3] / 1e3 if r[3] > 1e4 else r[3], r[4]))
PY
tail -2 $OUT/stats.log | cut -c1-200
cat $REPO/gpurun_out/kstats_$TAG.txt
rm -rf $OUT/stats
