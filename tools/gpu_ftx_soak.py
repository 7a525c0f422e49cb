#!/usr/bin/env python3
"""Soak of the pipelined full-refresh path (two lanes, column-sliced pipeline, one-kernel path below its threshold): random batch
sizes and offsets into one position set, several calls in flight, EVERY output compared with the one-kernel path's scores.
   python tools/gpu_ftx_soak.py [seconds]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import stormphrax_amd as sp

    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    total = 300000
    rng = np.random.default_rng(17)
    report = {}
    for preset in ("tame", "realistic"):
        net = sp.Network(sp.synthetic_net_bytes(preset))
        pos = sp.random_positions(total, seed=99, min_ply=0, max_ply=200, dfrc_every=3)
        d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
        with sp.NnueState(net, device=0, max_batch=1 << 18, sliced_ft=False) as plain:
            d_want = torch.empty(total, dtype=torch.int32, device="cuda")
            for lo in range(0, total, 1 << 18):
                m = min(1 << 18, total - lo)
                plain.evaluate_once_device(d_pos[lo:].data_ptr(), m, d_want[lo:].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        st = sp.NnueState(net, device=0, max_batch=1 << 18)
        outs = [torch.empty(1 << 18, dtype=torch.int32, device="cuda") for _ in range(6)]
        calls = positions = mismatching_calls = 0
        sizes = {"tiny<=8192": 0, "one_kernel": 0, "pipeline_one_pass": 0, "pipeline_many_passes": 0}
        t0 = time.time()
        while time.time() - t0 < seconds / 2:
            pending = []
            for o in outs:
                n = int(rng.choice([rng.integers(1, 8193), rng.integers(8193, 12288), rng.integers(12288, 65537), rng.integers(65537, 1 << 18)]))
                lo = int(rng.integers(0, total - n + 1))
                o[:n].fill_(-1)
                pending.append((o, n, lo))
            torch.cuda.synchronize()
            for o, n, lo in pending:
                st.evaluate_once_device_async(d_pos[lo:].data_ptr(), n, o.data_ptr())
                sizes["tiny<=8192" if n <= 8192 else "one_kernel" if n < 12288 else "pipeline_one_pass" if n <= 65536 else "pipeline_many_passes"] += 1
            st.synchronize()
            for o, n, lo in pending:
                calls += 1
                positions += n
                if not torch.equal(o[:n], d_want[lo:lo + n]):
                    mismatching_calls += 1
        st.close()
        report[preset] = {"seconds": round(time.time() - t0, 1), "calls": calls, "positions": positions, "calls_by_path": sizes,
                          "mismatching_calls": mismatching_calls}
    print(json.dumps(report, indent=1))
    return 0 if all(r["mismatching_calls"] == 0 for r in report.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
