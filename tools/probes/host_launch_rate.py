import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
print("loadavg", open("/proc/loadavg").read().strip(), "nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max ?", e)
t=time.perf_counter(); x=0
for i in range(5_000_000): x+=i
print("python 5M loop s", time.perf_counter()-t)
import torch
import stormphrax_amd as sp
from stormphrax_amd import _lib
lib=_lib.load()
pos=sp.random_positions(65536, seed=1)
d_pos=torch.from_numpy(pos.view(np.uint8).reshape(-1,32).copy()).cuda()
d_out=torch.empty(65536,dtype=torch.int32,device='cuda')
st=sp.NnueState(sp.Network(sp.synthetic_net_bytes("tame")), device=0, max_batch=65536)
for prof in (False, True, False, True):
    for rep in range(2):
        if prof: st.profile_begin(200)
        for i in range(30): st.evaluate_once_device_async(d_pos.data_ptr(), 65536, d_out.data_ptr())
        st.synchronize()
        t0=time.perf_counter()
        for i in range(200): st.evaluate_once_device_async(d_pos.data_ptr(), 65536, d_out.data_ptr())
        t1=time.perf_counter(); st.synchronize(); t2=time.perf_counter()
        if prof: st.profile_end()
        print("profiled" if prof else "plain   ", "issue 200 async calls: host %.1f us/call, total %.1f us/step -> %.3e evals/s" % ((t1-t0)/200*1e6, (t2-t0)/200*1e6, 65536*200/(t2-t0)))
# a trivial launch-rate probe
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    a=torch.zeros(64,device='cuda')
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(2000): a.add_(1)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("torch tiny kernel: host %.2f us/launch, total %.2f us/launch" % ((t1-t0)/2000*1e6, (t2-t0)/2000*1e6))
