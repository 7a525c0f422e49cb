#!/usr/bin/env python3
"""Column-sliced gather: when did each of its 256 workgroups start and end? Alone on the stream and pipelined (its preparation and the
other lane's kernels running beside it). Uniform slow-down or a tail of late / slow workgroups?"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fit_cost_model(plan, dur):
    """Per CU slot: groups, row quartets (steps) and segments of its planned range; least squares of the slots' mean workgroup time
    against them: what does a group / a step / a segment (slab load) cost?"""
    n_seg, n_groups = int(plan[32]), int(plan[33])
    bin_start = plan[256:256 + 1280].astype(np.int64)
    bucket_start = plan[256 + 1280:256 + 1280 + 17].astype(np.int64)
    # quartets of every group: the bin of its last member (sorted positions 8 G .. 8 G + 7; holes pad a bucket's last group)
    pos_bin = np.zeros(int(bucket_start[16]), dtype=np.int64)
    for b in range(16):
        for k in range(80):
            lo = bin_start[b * 80 + k]
            hi = bin_start[b * 80 + k + 1] if k < 79 else bucket_start[b] + (bucket_start[b + 1] - bucket_start[b])
            hi = min(hi, bucket_start[b + 1])
            pos_bin[lo:hi] = k + 1
        # holes at the end of the bucket inherit the last member's length
        end = bucket_start[b + 1]
        last = pos_bin[bucket_start[b]:end]
        if len(last):
            nz = np.nonzero(last)[0]
            if len(nz):
                last[nz[-1] + 1:] = last[nz[-1]]
    group_q = pos_bin.reshape(-1, 8).max(axis=1)
    rows = []
    for c in range(32):
        first, end = int(plan[c]), int(plan[c + 1]) if c < 31 else n_seg
        segs = [(int(plan[64 + 3 * k]), int(plan[64 + 3 * k + 1]), int(plan[64 + 3 * k + 2])) for k in range(first, end)]
        groups = sum(e - s for _, s, e in segs)
        steps = int(sum(group_q[s:e].sum() for _, s, e in segs))
        rows.append((groups, steps, len(segs), float(dur[8 * c:8 * c + 8].mean())))
    a = np.array([[g, q, sgm, 1.0] for g, q, sgm, _ in rows])
    y = np.array([t for *_, t in rows])
    coef, *_ = np.linalg.lstsq(a, y, rcond=None)
    pred = a @ coef
    return {"us_per_group": float(coef[0]), "us_per_step": float(coef[1]), "us_per_segment": float(coef[2]), "us_constant": float(coef[3]),
            "rms_residual_us": float(np.sqrt(((pred - y) ** 2).mean())),
            "per_cu_slot": [{"groups": g, "steps": q, "segments": sgm, "mean_us": round(t, 1)} for g, q, sgm, t in rows]}


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
            plan = np.zeros(256 + 1280 + 17, dtype=np.uint32)
            _lib.check(lib.spx_debug_ftx_plan(st._h, slot, plan.ctypes.data))
            fit = fit_cost_model(plan, dur)
            rows.append({"slot": slot, "cost_model_fit": fit, "kernel_us": float((end.max() - t0) / 100.0),
                         "start_spread_us": float((start.max() - t0) / 100.0),
                         "workgroup_us": {"min": float(dur.min()), "median": float(np.median(dur)), "p90": float(np.percentile(dur, 90)),
                                          "max": float(dur.max())},
                         "end_spread_us": float((end.max() - end.min()) / 100.0),
                         "per_xcd_median_us": [float(np.median(dur[x::8])) for x in range(8)],
                         # workgroup b = (XCD b % 8, CU slot b / 8): a slot's 8 workgroups walk the same planned range of groups
                         "per_cu_slot_mean_us": [round(float(dur[8 * c:8 * c + 8].mean()), 1) for c in range(32)],
                         "per_cu_slot_spread_over_xcds_us": [round(float(dur[8 * c:8 * c + 8].max() - dur[8 * c:8 * c + 8].min()), 1) for c in range(32)]})
        report[mode] = rows
        st.close()
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
