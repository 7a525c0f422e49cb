# rocprofv3 kernel stats of an arbitrary bench invocation: bash tools/gpu_stats.sh <tag> <bench args...>
# Per kernel AND launch grid: the default bench command also launches spx_ft_kernel for its secondary legs (the rebuild
# passes of the incremental leg: small grids; the wide-row and gather-ceiling legs: the headline's grid, other conditions),
# so one average per kernel NAME would mix them. The headline's launches are the group with the full grid and the most calls.
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python3 - "$@" <<PY
import glob, sqlite3, sys
print("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline " + " ".join(sys.argv[1:]))
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    print("# per kernel name (rocprofv3's own top_kernels view)")
    for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-60s calls %5d total %10.1f us avg %9.2f us  %5.1f%%" % (r[0][:60], r[1], r[2], r[3], r[4]))
    print("# per kernel name and grid (work-items): launches of one kernel under different conditions kept apart")
    q = "select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%spx%' group by name, grid_x order by sum(duration) desc"
    for name, grid, calls, avg, lo, hi in c.execute(q):
        print("%-52s grid %9d calls %5d avg %9.2f us  min %9.2f  max %9.2f" % (name.replace("spx::", "")[:52], grid, calls, avg / 1e3, lo / 1e3, hi / 1e3))
PY
