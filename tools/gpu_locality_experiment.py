#!/usr/bin/env python3
"""What would a finer ORDER of the batch buy the feature-transformer kernel? The kernel is bound by the L2-miss path (its
fabric traffic runs at ~5 TB/s); rows requested are the same whatever the order, so the FT kernel's time over permutations
of one batch isolates L2 locality. Orders tried: as generated; sorted by pawn structure; by the whole piece placement; by
material; and two floors (one position repeated; 1 024 distinct positions tiled). Product FT kernel alone on the stream
(spx_debug_gather_probe variant -1), medians over rounds."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pawn_masks(pos):
    """(white pawns, black pawns) bitboards of every record: nibble i of `pieces` = piece on the i-th set bit of occupancy
    (piece code: colour << 3 | type, pawn = 0; PackedBoard of the reference's marlinformat)."""
    occ = pos["occupancy"].copy()
    nib = pos["pieces"]
    wp = np.zeros(len(pos), dtype=np.uint64)
    bp = np.zeros(len(pos), dtype=np.uint64)
    for i in range(32):
        low = occ & (~occ + np.uint64(1))  # lowest set bit (0 when exhausted)
        code = (nib[:, i // 2] >> (4 * (i % 2))) & 0xF
        alive = low != 0
        wp |= np.where(alive & (code == 0), low, np.uint64(0))
        bp |= np.where(alive & (code == 8), low, np.uint64(0))
        occ &= occ - np.uint64(1) * alive.astype(np.uint64)
    return wp, bp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--preset", default="tame")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    import stormphrax_amd as sp

    net = sp.Network(sp.synthetic_net_bytes(args.preset))
    st = sp.NnueState(net, device=0, max_batch=args.batch)
    pos = sp.random_positions(args.batch, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
    wp, bp = pawn_masks(pos)
    raw = pos.view(np.uint8).reshape(-1, 32)
    npieces = np.array([bin(int(o)).count("1") for o in pos["occupancy"]])
    orders = {
        "as generated": np.arange(args.batch),
        "sorted by pawn structure (white pawns, black pawns)": np.lexsort((bp, wp)),
        "sorted by white pawns only": np.argsort(wp, kind="stable"),
        "sorted by occupancy, then the piece nibbles": np.lexsort(tuple(raw[:, 8 + k] for k in range(15, -1, -1)) + (pos["occupancy"],)),
        "sorted by piece count": np.argsort(npieces, kind="stable"),
        "one position repeated (floor: every row hits)": np.zeros(args.batch, dtype=np.int64),
        "1 024 distinct positions tiled": np.arange(args.batch) % 1024,
        "4 096 distinct positions tiled": np.arange(args.batch) % 4096,
    }
    bufs = {k: torch.from_numpy(np.ascontiguousarray(raw[o])).cuda() for k, o in orders.items()}
    times = {k: [] for k in orders}
    for _ in range(args.rounds):
        for k, d in bufs.items():
            _, ms, _ = st.gather_probe(d.data_ptr(), args.batch, -1, args.iters)
            times[k].append(ms)
    base = float(np.median(times["as generated"]))
    rows = []
    for k in orders:
        m = float(np.median(times[k]))
        w, c, t = st.count_rows(pos[orders[k]])
        rows.append({"order": k, "ft_kernel_us": m * 1e3, "vs_as_generated": m / base,
                     "requested_bytes": int(2048 * w + 1024 * (c + t))})
        print("%-60s %8.1f us  x%.3f  (%.2f GB requested)" % (k, m * 1e3, m / base, rows[-1]["requested_bytes"] / 1e9))
    if args.out:
        open(args.out, "w").write(json.dumps({"batch": args.batch, "preset": args.preset, "rows": rows}, indent=1) + "\n")


if __name__ == "__main__":
    main()
