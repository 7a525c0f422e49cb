# GPU occupancy of the device self-play: union of the kernel intervals over the middle 60 % of the run, per-kernel totals
# usage: bash tools/gpu_selfplay_busy.sh [games] [target]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sp_busy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPX_OPTIONS=selfplay_trace=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $REPO/tools/spx_selfplay.py --games ${1:-4096} --target ${2:-8192} > $OUT/run.log 2>&1
grep "spx_selfplay\]" $OUT/run.log
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/trace/*.db"):
    c = sqlite3.connect(f).cursor()
    rows = [r for r in c.execute("select name, start, end from kernels order by start") if "spx" in r[0]]
    t0, t1 = rows[0][1], rows[-1][2]
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    mid = [r for r in rows if r[1] >= lo and r[2] <= hi]
    busy, cur_s, cur_e = 0, None, None
    for n, s, e in mid:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("window %.1f ms, some kernel running %.1f %% of it" % ((hi - lo) / 1e6, 100.0 * busy / (hi - lo)))
    tot = {}
    for n, s, e in mid:
        k = n.replace("spx::", "").split("(")[0][:44]
        a = tot.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
    for k, (d, cnt) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print("  %-46s %7.1f %% of the window (sum of durations), %5d launches, avg %8.1f us" % (k, 100.0 * d / (hi - lo), cnt, d / cnt / 1e3))
PY
rm -rf $OUT/trace
