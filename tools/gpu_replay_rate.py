#!/usr/bin/env python3
"""BASELINE config 3 rate: the recorded 65 536-EVAL reference trace replayed (a) natively, level by level on device-resident
buffers (spx_acc_replay_tree), (b) level by level from the Python harness with host buffers (round-1 path)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormphrax_amd as sp  # noqa: E402
from stormphrax_amd.trace import Trace, replay, replay_native  # noqa: E402

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trace_startpos_tame_64k.txt.gz")
trace = Trace(path)
st = sp.NnueState(sp.Network.synthetic("tame"), device=0, max_batch=65536)
pos = trace.positions()
best, wall = 1e9, 1e9
for _ in range(5):
    t0 = time.perf_counter()
    got, want, ms = replay_native(st, trace, pos)
    wall = min(wall, (time.perf_counter() - t0) * 1e3)
    best = min(best, ms)
    assert np.array_equal(got, want)
t0 = time.perf_counter()
got, want, _ = replay(st, trace, pos)
harness_ms = (time.perf_counter() - t0) * 1e3
assert np.array_equal(got, want)
work = (trace.n_nodes - 1) + len(want)
print(json.dumps({"trace": os.path.basename(path), "updates": trace.n_nodes - 1, "evals": len(want),
                  "native_device_ms": best, "native_call_ms_incl_upload": wall, "python_harness_ms": harness_ms,
                  "native_updates_plus_evals_per_s": work / best * 1e3, "every_eval_equals_reference": True}))
