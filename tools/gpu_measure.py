#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (run on the GPU box): PCIe-inclusive rate of the host-buffer entry
point, incremental-path update rate (BASELINE config 3 shape), small-batch latency."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import stormphrax_amd as sp  # noqa: E402
from stormphrax_amd.trace import Trace  # noqa: E402

out = {}
net = sp.Network.synthetic("tame")
N = 65536
state = sp.NnueState(net, device=0, max_batch=N)
pos = sp.random_positions(N, seed=20260927)

# 1) host buffers in, host buffers out (H2D + kernels + D2H + sync per call)
for _ in range(3):
    state.evaluate_once(pos)
t0 = time.perf_counter()
reps = 30
for _ in range(reps):
    state.evaluate_once(pos)
dt = time.perf_counter() - t0
out["host_buffer_evals_per_s"] = N * reps / dt

# 1b) a host buffer of 16 chunks: the library pipelines the chunks over its two lanes (copies overlap evaluation)
big = np.tile(pos, 16)
state.evaluate_once(big)
t0 = time.perf_counter()
for _ in range(3):
    state.evaluate_once(big)
out["host_buffer_16_chunks_evals_per_s"] = len(big) * 3 / (time.perf_counter() - t0)

# 2) small batches (latency of one synchronous call)
for n in (1, 64, 1024, 4096, 8192):
    for _ in range(5):
        state.evaluate_once(pos[:n])
    t0 = time.perf_counter()
    for _ in range(200):
        state.evaluate_once(pos[:n])
    out[f"sync_call_us_batch_{n}"] = (time.perf_counter() - t0) / 200 * 1e6

# 3) incremental path: G concurrent games, one update batch + one eval batch per ply, device-resident inputs
G, PLIES = 32768, 24
rng = np.random.default_rng(1)
games = sp.random_positions(G, seed=7, min_ply=6, max_ply=40)
from stormphrax_amd import _lib  # noqa: E402

lib = _lib.load()
# pre-generate PLIES successor boards per game on the host (random legal moves via the chess core)
boards = [games]
import ctypes  # noqa: E402

def random_successors(cur, ply=[0]):
    ply[0] += 1
    nxt, moved = sp.random_successors(cur, seed=1000 + ply[0])  # C++ move generation; games without a move stay put
    nxt[~moved] = cur[~moved]
    return nxt

SUB = 2048  # host move generation through ctypes is slow: build plies for a subset and tile it
sub = games[:SUB]
chain = [sub]
for _ in range(PLIES):
    chain.append(random_successors(chain[-1]))
reps_tile = G // SUB
state.reserve_slots(2 * G)
d_boards = [torch.from_numpy(np.tile(b, reps_tile).view(np.uint8).reshape(-1, 32)).cuda() for b in chain]
slots_a = torch.arange(G, dtype=torch.int32, device="cuda")
slots_b = slots_a + G
d_out = torch.empty(G, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
h = state._h
_lib.check(lib.spx_acc_refresh_device(h, d_boards[0].data_ptr(), slots_a.data_ptr(), G, stream))
torch.cuda.synchronize()
def run(evaluate):
    _lib.check(lib.spx_acc_refresh_device(h, d_boards[0].data_ptr(), slots_a.data_ptr(), G, stream))
    torch.cuda.synchronize()
    cur, nxt = slots_a, slots_b
    for ply in range(1, PLIES + 1):
        _lib.check(lib.spx_acc_update_device(h, cur.data_ptr(), nxt.data_ptr(), d_boards[ply].data_ptr(), G, stream))
        if evaluate:
            _lib.check(lib.spx_acc_eval_device(h, nxt.data_ptr(), G, d_out.data_ptr(), stream))
        cur, nxt = nxt, cur
run(True)
torch.cuda.synchronize()
for evaluate in (False, True):
    _lib.check(lib.spx_acc_refresh_device(h, d_boards[0].data_ptr(), slots_a.data_ptr(), G, stream))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cur, nxt = slots_a, slots_b
    for ply in range(1, PLIES + 1):
        _lib.check(lib.spx_acc_update_device(h, cur.data_ptr(), nxt.data_ptr(), d_boards[ply].data_ptr(), G, stream))
        if evaluate:
            _lib.check(lib.spx_acc_eval_device(h, nxt.data_ptr(), G, d_out.data_ptr(), stream))
        cur, nxt = nxt, cur
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["incremental_updates_per_s" + ("_with_eval" if evaluate else "")] = G * PLIES / dt
# parity spot check of the last ply against a full refresh
full = torch.empty(G, dtype=torch.int32, device="cuda")
state.evaluate_once_device(d_boards[PLIES].data_ptr(), G, full.data_ptr(), stream)
torch.cuda.synchronize()
out["incremental_equals_full_refresh"] = bool(torch.equal(full, d_out))
# 4) BASELINE config 3: the recorded 65 536-EVAL reference trace, replayed level by level through the arena
#    (host-buffer entry points; every EVAL is checked against the value the reference recorded)
from stormphrax_amd.trace import replay  # noqa: E402

trace = Trace(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                           "trace_startpos_tame_64k.txt.gz"))
tpos = trace.positions()
replay(state, trace, tpos)
t0 = time.perf_counter()
for _ in range(5):
    got, ref_inc, _ref_once = replay(state, trace, tpos)
dt = (time.perf_counter() - t0) / 5
out["trace_64k_replay_ms"] = dt * 1e3
out["trace_64k_updates_plus_evals_per_s"] = (trace.n_nodes - 1 + len(trace.evals)) / dt
out["trace_64k_matches_reference"] = bool(np.array_equal(got, ref_inc))
# 5) viriformat -> records: host replay (validating) vs device replay (incl. the copies both ways)
blob = b"".join(sp.viri_random_game(7000 + s, plies=160, dfrc=(s % 4 == 0)) for s in range(2000))
t0 = time.perf_counter()
host_pos, _ = sp.viri_expand(blob)
out["viri_expand_host_positions_per_s"] = len(host_pos) / (time.perf_counter() - t0)
state.viri_expand(blob)
t0 = time.perf_counter()
for _ in range(5):
    dev_pos, _, _ = state.viri_expand(blob)
out["viri_expand_device_positions_per_s"] = len(dev_pos) * 5 / (time.perf_counter() - t0)
out["viri_expand_identical"] = bool(dev_pos.tobytes() == host_pos.tobytes())
print(json.dumps(out))
