import sys, time, os
sys.path.insert(0, '.')
import numpy as np, stormphrax_amd as sp
net = sp.Network.synthetic("tame")
st = sp.NnueState(net, device=0, max_batch=65536)
pos = sp.random_positions(8192, seed=20260927)
out = {}
for n in (1, 64, 256, 512, 1024, 2048):
    for _ in range(20): st.evaluate_once(pos[:n])
    t0 = time.perf_counter()
    for _ in range(500): st.evaluate_once(pos[:n])
    out[n] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
print(os.environ.get("SPX_TINY_BATCH_MAX", "default"), out)
