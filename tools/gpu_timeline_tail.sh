# kernel timeline of the LAST steps of a bench run (the timed region of a default run, behind its settle phase): bash tools/gpu_timeline_tail.sh [bench args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/timeline_tail
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/bench.py --no-cpu-baseline --no-wide --no-secondary "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | python $REPO/tools/bench_brief.py /dev/stdin
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = [r for r in c.execute(f"select name, start, end, {q} from kernels order by start") if "spx" in r[0]]
    rows = rows[-110:-40]
    t0 = rows[0][1]
    for n, s, e, qid in rows:
        print("queue %-3s %-30s start %9.2f  end %9.2f  dur %7.2f" % (qid, n.replace("spx::", "").replace("void ", "")[:30], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf $OUT/*.db
