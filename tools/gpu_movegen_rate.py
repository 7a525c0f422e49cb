#!/usr/bin/env python3
"""Throughput of the device move generator (spx_movegen_device): positions/s and children/s on a resident batch."""
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

N = 65536
st = sp.NnueState(sp.Network.synthetic("tame"), device=0, max_batch=N)
pos = sp.random_positions(N, seed=20260927)
cap = N * 64
d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
d_children = torch.empty((cap, 32), dtype=torch.uint8, device="cuda")
d_moves = torch.empty(cap, dtype=torch.int16, device="cuda")
d_parents = torch.empty(cap, dtype=torch.int32, device="cuda")
d_first = torch.empty(N, dtype=torch.int32, device="cuda")
d_count = torch.empty(N, dtype=torch.int32, device="cuda")
d_check = torch.empty(N, dtype=torch.uint8, device="cuda")
d_total = torch.zeros(1, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
lib = _lib.load()


def run():
    _lib.check(lib.spx_movegen_device(st._h, d_pos.data_ptr(), N, None, d_children.data_ptr(), d_moves.data_ptr(),
                                      d_parents.data_ptr(), d_first.data_ptr(), d_count.data_ptr(), d_check.data_ptr(),
                                      cap, d_total.data_ptr(), stream))


for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 50
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
total = int(d_total.item())
print(json.dumps({"positions": N, "children": total, "ms": dt * 1e3, "positions_per_s": N / dt, "children_per_s": total / dt,
                  "bytes_written_per_s": total * 38 / dt}))
