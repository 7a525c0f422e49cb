"""Worker of tests/test_gpu_configs.py::test_config4_hbm_filling_batch (BASELINE configs[4], the HBM-filling batch).

The batch is a device-side tiling of 131 072 distinct positions, so properties that need no CPU pass over billions of
positions pin the result: every tile must repeat tile 0's scores (independence of position order and of the internal chunk
boundaries - tiles straddle the 4 Mi-position chunks) and tile 0 is checked against the CPU oracle on a sample."""
import json
import os
import sys

import numpy as np
import torch  # first: its HIP runtime must be the one the process initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stormphrax_amd as sp  # noqa: E402
from conftest import Oracle  # noqa: E402


def main():
    blob = sp.synthetic_net_bytes("tame")
    free, total = torch.cuda.mem_get_info(0)
    distinct = 1 << 17
    n = int(free * 0.90 - 12e9) // 36 // distinct * distinct  # room for the net, 2 x ~4.6 GB of chunk scratch, torch itself
    assert n >= 64 * distinct, f"only {free / 1e9:.0f} GB free"
    base = sp.random_positions(distinct, seed=99, min_ply=8, max_ply=120, dfrc_every=4)
    d_base = torch.from_numpy(base.view(np.uint8).reshape(-1, 32)).cuda()
    d_pos = d_base.repeat(n // distinct, 1)
    d_out = torch.empty(n, dtype=torch.int32, device="cuda")
    st = sp.NnueState(sp.Network(blob), device=0, max_batch=n)
    assert st.scratch_batch < n
    import time

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.evaluate_once_device_async(d_pos.data_ptr(), n, d_out.data_ptr())
    st.synchronize()
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0  # ONE call over the whole resident batch (first call: includes the lanes' allocation)
    in_use = total - torch.cuda.mem_get_info(0)[0]
    tiles = d_out.view(n // distinct, distinct)
    same = True
    for lo in range(0, tiles.shape[0], 4096):  # in slabs: a full-size comparison mask would not fit beside the batch
        same = same and bool((tiles[lo:lo + 4096] == tiles[0]).all())
    sample = np.arange(0, distinct, distinct // 4096)
    oracle = Oracle()
    oracle.use(blob, "tame")
    mail, stm = sp.positions_to_mailboxes(base[sample])
    exact = bool(np.array_equal(tiles[0].cpu().numpy()[sample], oracle.eval_mailboxes(mail, stm)))
    line = {"config": "BASELINE configs[4]: the largest position batch that fits the HBM, one spx_eval_full_device_async call",
            "positions": n, "chunks": -(-n // st.scratch_batch), "in_use_gb": in_use / 1e9, "total_gb": total / 1e9,
            "seconds": seconds, "evals_per_sec": n / seconds, "tiles_identical": same, "oracle_sample_exact": exact}
    print(json.dumps(line))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):  # (the GPU box: kept as a profile of the round)
        with open(os.path.join(out_dir, "config5_hbm_filling.json"), "w") as f:
            f.write(json.dumps(line, indent=1) + "\n")
    st.close()


if __name__ == "__main__":
    main()
