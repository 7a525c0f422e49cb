# rocprofv3 kernel stats of an arbitrary bench invocation: bash tools/gpu_stats.sh <tag> <bench args...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-60s calls %5d total %10.1f us avg %9.2f us  %5.1f%%" % (r[0][:60], r[1], r[2], r[3], r[4]))
PY
