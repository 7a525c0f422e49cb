import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import stormphrax_amd as sp
from stormphrax_amd.trace import Trace, replay_native
net = sp.Network(sp.synthetic_net_bytes("tame"))
for name in ("trace_search_startpos_tame_64k.txt.gz", "trace_startpos_tame_64k.txt.gz"):
    trace = Trace(os.path.join("tests", "golden", name)); pos = trace.positions()
    for seg in (4, 8, 12, 16, 24, 32, 64):
        with sp.NnueState(net, device=0, max_batch=65536, options={"replay_segment": seg}) as st:
            best = 1e9
            for _ in range(4):
                got, want, ms = replay_native(st, trace, pos); best = min(best, ms)
            print(name[:28], "segment", seg, "ms %.3f" % best, "exact", bool(np.array_equal(got, want)), "rate %.3e" % ((trace.n_nodes - 1 + len(want)) / best * 1e3))
