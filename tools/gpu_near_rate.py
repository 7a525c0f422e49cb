"""What near-compact piece-square rows buy (GPU box): the tame synthetic net with K weights per piece-square row pushed
outside i8 (a stand-in for a trained net whose rows almost fit) evaluated (a) with the rows served as 1 KiB copies +
remainders (default), (b) with SPX_OPTIONS=near_rows=0: every such row fetched as its 2 KiB i16 row. Same scores, checked against
each other and, on a sample, against the CPU oracle (test infrastructure).

    gpurun -- 'python tools/gpu_near_rate.py > gpurun_out/near_rate.json'
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def outlier_net(sp, per_row, seed=5):
    blob = np.array(sp.synthetic_net_bytes("tame"), copy=True)
    psq = blob[64 : 64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024)
    rng = np.random.default_rng(seed)
    for r in range(11264):
        cols = rng.choice(1024, size=per_row, replace=False)
        psq[r, cols] = rng.choice(np.array([-300, -180, 150, 220, 400], dtype=np.int16), size=per_row)
    return blob


def rate(sp, torch, blob, d_pos, n, steps=200):
    net = sp.Network(blob)
    with sp.NnueState(net, max_batch=n) as st:
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        for _ in range(600):  # settle
            st.evaluate_once_device_async(d_pos.data_ptr(), n, out.data_ptr())
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st.evaluate_once_device_async(d_pos.data_ptr(), n, out.data_ptr())
        st.synchronize()
        dt = time.perf_counter() - t0
        return n * steps / dt, out.cpu().numpy(), st.compact_psq_rows, st.near_psq_rows


def main():
    import torch

    import stormphrax_amd as sp

    n = 65536
    pos = sp.random_positions(n, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
    d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
    report = {"positions_per_step": n, "cases": []}
    base, ref_scores, c, q = rate(sp, torch, sp.synthetic_net_bytes("tame"), d_pos, n)
    report["cases"].append({"net": "tame (every row fits i8)", "evals_per_s": base, "compact_rows": c, "near_rows": q})
    oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "libspx_oracle.so"))
    oracle.spxo_init.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    oracle.spxo_eval_mailboxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    mail, stm = sp.positions_to_mailboxes(pos[:2048])
    for per_row in (2, 8, 16, 32):
        blob = outlier_net(sp, per_row)
        os.environ.pop("SPX_OPTIONS", None)
        near, s_near, c1, q1 = rate(sp, torch, blob, d_pos, n)
        os.environ["SPX_OPTIONS"] = "near_rows=0"
        wide, s_wide, c2, q2 = rate(sp, torch, blob, d_pos, n)
        os.environ.pop("SPX_OPTIONS", None)
        assert oracle.spxo_init(blob.ctypes.data, blob.size) == 0
        want = np.empty(2048, dtype=np.int32)
        oracle.spxo_eval_mailboxes(mail.ctypes.data, stm.ctypes.data, 2048, want.ctypes.data)
        report["cases"].append({
            "net": f"tame + {per_row} weights outside i8 in every piece-square row",
            "near_path_evals_per_s": near, "near_rows": q1, "compact_rows": c1,
            "wide_rows_evals_per_s": wide, "near_rows_with_near_rows_0": q2,
            "speedup": near / wide, "identical_scores": bool(np.array_equal(s_near, s_wide)),
            "oracle_sample_ok": bool(np.array_equal(s_near[:2048], want)),
        })
    print(json.dumps(report))


if __name__ == "__main__":
    main()
