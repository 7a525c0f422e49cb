# kernel timeline of the incremental ply at a launch-bound size: durations and gaps between consecutive kernels
# usage: bash tools/gpu_gaps.sh <games>
G=${1:-4096}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/gaps_$G
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/bench.py --mode incremental --batch $G --steps 200 --warmup 20 --no-cpu-baseline > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(f).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    name = [t for t in tabs if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tabs) else None
    print("tables:", [t for t in tabs if "kernel" in t][:8])
    rows = list(c.execute("select name, start, end from kernels order by start"))
    rows = [r for r in rows if "spx" in r[0]]
    rows = rows[len(rows)//2:len(rows)//2+16]
    prev = None
    for n, s, e in rows:
        print("%-40s dur %7.2f us  gap %7.2f us" % (n[:40], (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3))
        prev = e
PY
