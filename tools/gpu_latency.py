import sys, time, os
sys.path.insert(0, '.')
import numpy as np, stormphrax_amd as sp
net = sp.Network.synthetic("tame")
st = sp.NnueState(net, device=0, max_batch=65536)
pos = sp.random_positions(8192, seed=20260927)
out = {}
for n in (1, 64, 256, 512, 1024, 2048, 4096, 8192):
    for _ in range(20): st.evaluate_once(pos[:n])
    t0 = time.perf_counter()
    for _ in range(500): st.evaluate_once(pos[:n])
    out[n] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
print(os.environ.get("SPX_OPTIONS", "default"), out)
# push + evaluate of n nodes (spx_acc_update_eval, the search's own step) and evaluate of materialised slots
st.reserve_slots(8192)
slots = np.arange(2048, dtype=np.uint32)
st.reset(pos[:2048], slots)
nxt, moved = sp.random_successors(pos[:2048], seed=5)
out2 = {}
for n in (1, 64, 1024, 2048):
    for _ in range(20): st.update_evaluate(slots[:n], slots[:n] + 4096, nxt[:n])
    t0 = time.perf_counter()
    for _ in range(500): st.update_evaluate(slots[:n], slots[:n] + 4096, nxt[:n])
    a = (time.perf_counter() - t0) / 500 * 1e6
    t0 = time.perf_counter()
    for _ in range(500): st.evaluate(slots[:n])
    out2[n] = (round(a, 1), round((time.perf_counter() - t0) / 500 * 1e6, 1))
print("update+eval / eval-of-slots us:", out2)
