# Per-launch durations of a native tree replay of the alpha-beta search trace (rocprofv3 --kernel-trace): the chain launches of the
# path walk (one per round of heavy paths) and the evaluation kernels behind them
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/replay_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/rp.py <<PY
import os, sys
sys.path.insert(0, "$REPO")
import numpy as np, stormphrax_amd as sp
from stormphrax_amd.trace import Trace, replay_native
t = Trace("$REPO/tests/golden/trace_search_startpos_tame_64k.txt.gz")
st = sp.NnueState(sp.Network(sp.synthetic_net_bytes("tame")), device=0, max_batch=65536)
pos = t.positions()
for _ in range(3):
    got, want, ms = replay_native(st, t, pos)
print(ms, np.array_equal(got, want))
PY
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python /tmp/rp.py > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/trace/*.db"):
    c = sqlite3.connect(f).cursor()
    rows = [r for r in c.execute("select name, start, end, grid_x from kernels order by start") if "spx" in r[0]]
    # the last replay's chain launches: the last run of chain launches without a gap of more than 1 ms
    allc = [r for r in rows if "chain" in r[0]]
    chains = [allc[-1]]
    for r in reversed(allc[:-1]):
        if chains[0][1] - r[2] > 1e6: break
        chains.insert(0, r)
    for n, s, e, g in chains:
        print("chain launch grid %8d  %8.1f us" % (g, (e - s) / 1e3))
    t0 = chains[0][1]; tail = [r for r in rows if r[1] >= t0]
    print("from the first chain launch to the last kernel of the replay: %.1f us; kernels after the chains:" % ((tail[-1][2] - t0) / 1e3))
    for n, s, e, g in tail[len(chains):]:
        print("   %-40s %8.1f us (starts at +%.1f)" % (n.replace("spx::", "")[:40], (e - s) / 1e3, (s - t0) / 1e3))
PY
rm -rf $OUT/trace
