#!/usr/bin/env python3
"""Throughput of the two incremental update paths on the same (parent -> child) records, device-resident operands:
board-diff kernel (spx_acc_update_eval_device) vs. the reference's own bookkeeping (observer deltas captured on the host,
spx_acc_update_observed_device)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import stormphrax_amd as sp  # noqa: E402
from stormphrax_amd import _lib  # noqa: E402

lib = _lib.load()
SUB, N = 4096, 65536
st = sp.NnueState(sp.Network.synthetic("tame"), device=0, max_batch=N)
base = sp.random_positions(SUB, seed=11, min_ply=6, max_ply=80, dfrc_every=4)
rng = np.random.default_rng(3)
children = np.zeros(SUB, dtype=sp.PACKED_DTYPE)
deltas = (_lib.MoveDelta * SUB)()
files = "abcdefgh"
for i in range(SUB):
    moves, _, _ = sp.legal_moves(base[i])
    if len(moves) == 0:
        base[i] = base[(i + 1) % SUB]
        moves, _, _ = sp.legal_moves(base[i])
    mv = int(moves[rng.integers(len(moves))])
    frm, to, promo, kind = mv & 63, (mv >> 6) & 63, (mv >> 12) & 3, mv >> 14
    uci = files[frm & 7] + str((frm >> 3) + 1) + files[to & 7] + str((to >> 3) + 1) + ("nbrq"[promo] if kind == 3 else "")
    rc = lib.spx_pos_apply_uci_observed(base[i:i + 1].ctypes.data, uci.encode(), children[i:i + 1].ctypes.data,
                                        ctypes.byref(deltas[i]))
    assert rc == 0, uci
reps = N // SUB
parents_h = np.tile(base, reps)
children_h = np.tile(children, reps)
delta_bytes = np.frombuffer(bytes(deltas), dtype=np.uint8).reshape(SUB, -1)
d_deltas = torch.from_numpy(np.tile(delta_bytes, (reps, 1)).copy()).cuda()
d_children = torch.from_numpy(children_h.view(np.uint8).reshape(-1, 32)).cuda()
st.reserve_slots(2 * N)
slots = np.arange(N, dtype=np.uint32)
st.reset(parents_h, slots)
d_par = torch.arange(N, dtype=torch.int32, device="cuda")
d_chi = d_par + N
d_out = torch.empty(N, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
out = {"records": N, "delta_bytes_per_record": int(delta_bytes.shape[1])}


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    return N * 50 / (time.perf_counter() - t0)


out["board_diff_updates_plus_evals_per_s"] = timed(lambda: _lib.check(lib.spx_acc_update_eval_device(
    st._h, d_par.data_ptr(), d_chi.data_ptr(), d_children.data_ptr(), N, d_out.data_ptr(), stream)))
a = d_out.clone()
out["observed_updates_plus_evals_per_s"] = timed(lambda: _lib.check(lib.spx_acc_update_observed_device(
    st._h, d_par.data_ptr(), d_chi.data_ptr(), d_children.data_ptr(), d_deltas.data_ptr(), N, d_out.data_ptr(), stream)))
out["identical"] = bool(torch.equal(a, d_out))
print(json.dumps(out))
