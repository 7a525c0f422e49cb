import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import stormphrax_amd as sp
from stormphrax_amd import _lib
lib = _lib.load()
lib.spx_debug_ft_gate.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
N = 65536
net = sp.Network.synthetic("tame")
pos = sp.random_positions(N, seed=20260927)
d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
def run(mode, steps=200):
    sts = [sp.NnueState(net, device=0, max_batch=N) for _ in range(2)]
    outs = [torch.empty(N, dtype=torch.int32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    if mode == "gated":
        lib.spx_debug_ft_gate(sts[0]._h, sts[1]._h)
    def step(k):
        i = k % 2 if mode != "single" else 0
        sts[i].evaluate_once_device(d_pos.data_ptr(), N, outs[i].data_ptr(), streams[i].cuda_stream)
    for k in range(20): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = outs[0].clone()
    print(mode, "%.3e evals/s" % (N * steps / dt), "%.4f ms/step" % (dt / steps * 1e3), bool(torch.equal(outs[0], outs[1])) if mode != "single" else "")
    for s in sts: s.close()
for m in ("single", "two_streams", "gated", "single", "gated"):
    run(m)
