"""Does the pipelined step rate hold under sustained load? Chunks of 10 async calls (as bench.py's settle loop issues them), a
synchronise after each, for ~6 s; per 25 chunks the rate, and rocm-smi's clocks / power sampled beside it."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import stormphrax_amd as sp
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pos = sp.random_positions(65536, seed=1)
d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32).copy()).cuda()
d_out = torch.empty(65536, dtype=torch.int32, device="cuda")
st = sp.NnueState(sp.Network(sp.synthetic_net_bytes("tame")), device=0, max_batch=65536)
st.evaluate_once_device_async(d_pos.data_ptr(), 65536, d_out.data_ptr()); st.synchronize()
plain_s = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
total_s = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
profiled = False
t_start = time.perf_counter()
while time.perf_counter() - t_start < total_s:
    if not profiled and time.perf_counter() - t_start >= plain_s:
        st.profile_begin(60000); profiled = True
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.25:
        for i in range(chunk): st.evaluate_once_device_async(d_pos.data_ptr(), 65536, d_out.data_ptr())
        st.synchronize(); n += chunk
    dt = time.perf_counter() - t0
    smi = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed 's/.*: //' | tr '\\n' ' '", shell=True, capture_output=True, text=True).stdout
    print(("prof " if profiled else "plain") + " t=%.2fs chunk %d: %.3e evals/s   %s" % (time.perf_counter() - t_start, chunk, 65536 * n / dt, smi.strip()))
