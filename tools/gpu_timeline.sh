# kernel timeline of the pipelined full-refresh steps (which kernels overlap, where the gathers wait): bash tools/gpu_timeline.sh [bench args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-settle --no-wide --no-secondary "$@" > $OUT/run.log 2>&1
python3 - <<PY > $REPO/gpurun_out/timeline.txt
import glob, sqlite3
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = [r for r in c.execute(f"select name, start, end, {q} from kernels order by start") if "spx" in r[0]]
    mid = len(rows) * 2 // 3
    while "gather" not in rows[mid][0] and "ft_kernel" not in rows[mid][0]:
        mid += 1
    rows = rows[mid:mid + 40]
    t0 = rows[0][1]
    print("# rocprofv3 --kernel-trace of: bench.py --steps 60 --warmup 10 --no-settle --no-wide --no-secondary $*  (a window of steady-state steps; us from the window's first kernel)")
    for n, s, e, qid in rows:
        print("queue %-3s %-34s start %9.2f  end %9.2f  dur %7.2f" % (qid, n.replace("spx::", "").replace("void ", "")[:34], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat $REPO/gpurun_out/timeline.txt
rm -rf $OUT/*.db
