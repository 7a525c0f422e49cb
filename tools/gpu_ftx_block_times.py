#!/usr/bin/env python3
"""Column-sliced gather: when did each of its 256 workgroups start and end? Alone on the stream and pipelined (its preparation and the
other lane's kernels running beside it). Uniform slow-down or a tail of late / slow workgroups?"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import stormphrax_amd as sp
    from stormphrax_amd import _lib

    lib = _lib.load()
    batch = 65536
    net = sp.Network(sp.synthetic_net_bytes("tame"))
    pos = sp.random_positions(batch, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
    d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
    outs = [torch.empty(batch, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    report = {}
    for mode in ("stream_ordered", "pipelined"):
        st = sp.NnueState(net, device=0, max_batch=batch, sliced_ft=True)
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(60):
            if mode == "pipelined":
                st.evaluate_once_device_async(d_pos.data_ptr(), batch, outs[i & 1].data_ptr())
            else:
                st.evaluate_once_device(d_pos.data_ptr(), batch, outs[i & 1].data_ptr(), stream)
        st.synchronize()
        torch.cuda.synchronize()
        rows = []
        for slot in ((-1,) if mode == "stream_ordered" else (0, 1)):
            t = np.zeros(512, dtype=np.uint64)
            _lib.check(lib.spx_debug_ftx_block_times(st._h, slot, t.ctypes.data))
            start, end = t[0::2].astype(np.int64), t[1::2].astype(np.int64)
            t0 = start.min()
            dur = (end - start) / 100.0  # us
            rows.append({"slot": slot, "kernel_us": float((end.max() - t0) / 100.0),
                         "start_spread_us": float((start.max() - t0) / 100.0),
                         "workgroup_us": {"min": float(dur.min()), "median": float(np.median(dur)), "p90": float(np.percentile(dur, 90)),
                                          "max": float(dur.max())},
                         "end_spread_us": float((end.max() - end.min()) / 100.0),
                         "per_xcd_median_us": [float(np.median(dur[x::8])) for x in range(8)]})
        report[mode] = rows
        st.close()
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
