# The serial chain of one self-play lane: per-stream kernel durations and the gaps between consecutive kernels
# usage: bash tools/gpu_selfplay_chain.sh [games] [target]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sp_chain
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $REPO/tools/spx_selfplay.py --games ${1:-4096} --target ${2:-32768} --dfrc > $OUT/run.log 2>&1
python3 - <<PY
import glob, sqlite3, collections
for f in glob.glob("$OUT/trace/*.db"):
    c = sqlite3.connect(f).cursor()
    rows = [r for r in c.execute("select name, start, end, stream_id, queue_id from kernels order by start") if "spx" in r[0]]
    t0, t1 = rows[0][1], rows[-1][2]
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    by = collections.defaultdict(list)
    for n, s, e, st, q in rows:
        if s >= lo and e <= hi: by[(st, q)].append((n.replace("spx::", "").split("(")[0].replace("void ", "")[:40], s, e))
    for key, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
        if len(ks) < 100: continue
        dur = collections.defaultdict(lambda: [0, 0]); gap = collections.defaultdict(lambda: [0, 0])
        for (n, s, e), nxt in zip(ks, ks[1:]):
            d = dur[n]; d[0] += e - s; d[1] += 1
            g = gap[n]; g[0] += max(0, nxt[1] - e); g[1] += 1
        span = ks[-1][2] - ks[0][1]
        print("stream %s queue %s: %d kernels over %.1f ms; kernels %.1f %%, gaps %.1f %%" % (key[0], key[1], len(ks), span / 1e6,
              100.0 * sum(d[0] for d in dur.values()) / span, 100.0 * sum(g[0] for g in gap.values()) / span))
        for n in sorted(dur, key=lambda k: -dur[k][0]):
            print("   %-42s %5d x avg %7.1f us (%.1f %% of the span), gap after it avg %6.1f us" % (n, dur[n][1], dur[n][0] / dur[n][1] / 1e3,
                  100.0 * dur[n][0] / span, gap[n][0] / max(1, gap[n][1]) / 1e3))
PY
rm -rf $OUT/trace
