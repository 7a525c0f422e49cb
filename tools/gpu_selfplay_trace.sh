# kernel timeline of a few plies of the device self-play (rocprofv3 kernel trace): name, start, duration, stream/queue
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sp_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/tools/spx_selfplay.py --games ${1:-4096} --target ${2:-4096} > $OUT/run.log 2>&1
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print(cols)
    rows = list(c.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
    rows = [r for r in rows if "spx" in r[0]]
    mid = len(rows) // 2
    t0 = rows[mid][1]
    for n, s, e, q, st in rows[mid:mid + 60]:
        print("%-34s start %9.1f us dur %8.1f us  q %s s %s" % (n.replace("spx::", "")[:34], (s - t0) / 1e3, (e - s) / 1e3, q, st))
PY
