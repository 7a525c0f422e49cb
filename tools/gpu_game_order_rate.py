#!/usr/bin/env python3
"""Full-refresh throughput on GAME-ORDERED positions (consecutive plies of self-play games, as a rescoring run sees them)
vs. the same positions shuffled: consecutive positions share most of their feature rows, so L1/L2 hit rates rise."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import stormphrax_amd as sp  # noqa: E402

net = sp.Network.synthetic("tame")
st = sp.NnueState(net, device=0, max_batch=4096 * 64)
path = "/tmp/spx_games.vf"
st.selfplay(4096, 6000, out_path=path, max_plies=200, temperature_cp=30, seed=3)
positions, games, bad = st.viri_expand(open(path, "rb").read())
os.remove(path)
N = 1 << 19
positions = positions[:N]
assert len(positions) == N
st2 = sp.NnueState(net, device=0, max_batch=N)
out = {"positions": N, "games": games}
rng = np.random.default_rng(0)
for name, batch in (("game_order", positions), ("shuffled", positions[rng.permutation(N)])):
    d_pos = torch.from_numpy(batch.view(np.uint8).reshape(-1, 32)).cuda()
    d_out = torch.empty(N, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        st2.evaluate_once_device(d_pos.data_ptr(), N, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st2.evaluate_once_device(d_pos.data_ptr(), N, d_out.data_ptr(), stream)
    torch.cuda.synchronize()
    out[name + "_evals_per_s"] = N * 20 / (time.perf_counter() - t0)
print(json.dumps(out))
