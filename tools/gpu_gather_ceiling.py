#!/usr/bin/env python3
"""Gather-ceiling probe on the bench batch: load-only replays of the full refresh's row fetches -
global_load_dwordx4 at 5 / 6 / 8 waves per SIMD, LDS-DMA rings - next to the product FT kernel, all in one process on the
same device buffers, interleaved over `--rounds` rounds. Writes one JSON object (stdout and --out)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--preset", default="tame")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--wide", action="store_true", help="context with SPX_CTX_WIDE_PSQ_ROWS")
    ap.add_argument("--variants", default=None, help="comma-separated variant ids (-1 = the product kernel); default all")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    import stormphrax_amd as sp

    net = sp.Network(sp.synthetic_net_bytes(args.preset))
    st = sp.NnueState(net, device=0, max_batch=args.batch, wide_psq_rows=args.wide)
    pos = sp.random_positions(args.batch, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)  # bench.py's rank-0 batch
    d_pos = torch.from_numpy(pos.view(np.uint8).reshape(-1, 32)).cuda()
    wide_rows, compact_rows, thr_rows = st.count_rows(pos)
    requested = 2048 * wide_rows + 1024 * (compact_rows + thr_rows)
    variants = [-1] + list(range(st.gather_probe_variants()))
    if args.variants:
        variants = [int(v) for v in args.variants.split(",")]
        if -1 not in variants:
            variants = [-1] + variants
    times = {v: [] for v in variants}
    names, sinks = {}, {}
    from stormphrax_amd._lib import SpxError

    for _ in range(args.rounds):
        for v in list(variants):
            try:
                name, ms, sink = st.gather_probe(d_pos.data_ptr(), args.batch, v, args.iters)
            except SpxError:  # (the column-sliced replays cover 1 KiB rows only: not for --wide contexts)
                variants.remove(v)
                times.pop(v, None)
                continue
            names[v], sinks[v] = name, sink
            times[v].append(ms)
    ft_us = float(np.median(times[-1])) * 1e3
    rows = []
    for v in variants:
        us = float(np.median(times[v])) * 1e3
        rows.append({"variant": v, "name": names[v], "us_per_launch": us, "min_us": min(times[v]) * 1e3,
                     "max_us": max(times[v]) * 1e3, "sink_checksum": sinks[v], "requested_gbs": requested / us / 1e3,
                     "ft_kernel_over_this": ft_us / us})
    probe_sinks = {sinks[v] for v in variants if v >= 0}
    best = min((r for r in rows if r["variant"] >= 0), key=lambda r: r["us_per_launch"])
    out = {"batch": args.batch, "preset": args.preset, "wide_psq_rows": args.wide,
           "rows_per_launch": {"psq_wide_2KiB": wide_rows, "psq_compact_1KiB": compact_rows, "threat_1KiB": thr_rows},
           "requested_bytes_per_launch": requested, "iters": args.iters, "rounds": args.rounds,
           "all_probe_variants_loaded_the_same_bytes": len(probe_sinks) == 1,
           "ft_kernel_us": ft_us, "best_probe": best["name"], "best_probe_us": best["us_per_launch"],
           "ft_kernel_frac_of_probe": best["us_per_launch"] / ft_us, "variants": rows}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
