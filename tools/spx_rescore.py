#!/usr/bin/env python3
"""Re-evaluate recorded training data on the GPU (needs an MI355X).

    python tools/spx_rescore.py <in.bin|in.vf> <out.bin> [--net file.nnue | --preset tame] [--batch 1048576]

Input: marlinformat `.bin` (32-byte PackedBoard records, src/datagen/marlinformat.h:32-84) or viriformat `.vf` game
streams (src/datagen/viriformat.cpp:28-63, expanded to one record per played move). Output: marlinformat records whose
`eval` field holds the raw network output of the position from WHITE's point of view (clamped to i16) - the convention of
the reference's own data files (Searcher::runDatagenSearch returns the white-relative score, src/search.cpp:237, and
Marlinformat / Viriformat store it, datagen.cpp:283-284) - everything else unchanged: the same 32-byte records
bullet/marlinflow-style trainers consume. --pov stm keeps the side-to-move-relative value instead."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormphrax_amd as sp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--net")
    ap.add_argument("--preset", default="tame")
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--filter", action="store_true",
                    help=".vf input: keep only the positions the reference's marlinformat output keeps (side to move not in "
                         "check, played move not a capture / en passant / queen promotion: datagen.cpp:254)")
    ap.add_argument("--pov", default="white", choices=["white", "stm"],
                    help="point of view of the written evals: white (the reference's data convention) or the side to move")
    ap.add_argument("--validate", action="store_true",
                    help=".vf input: replay the games on the host and check every move against the legal-move generator "
                         "(~5e5 positions/s) instead of the device replay, which trusts the stream")
    args = ap.parse_args()
    raw = open(args.src, "rb").read()
    net = sp.Network(open(args.net, "rb").read()) if args.net else sp.Network.synthetic(args.preset)
    if args.src.endswith(".vf"):
        if args.validate:
            positions, games, keep = sp.viri_expand(raw, with_filter=True)
        else:
            with sp.NnueState(net, device=args.device, max_batch=1) as expander:
                positions, games, bad, keep = expander.viri_expand(raw, with_filter=True)
            if bad:
                print(f"warning: {bad} games contain a move from a square without a piece of the side to move", file=sys.stderr)
        total = len(positions)
        if args.filter:
            positions = positions[keep]
        print(f"{games} games -> {total} positions" + (f", {len(positions)} after filtering" if args.filter else ""))
    else:
        positions = np.frombuffer(raw, dtype=sp.PACKED_DTYPE).copy()
    state = sp.NnueState(net, device=args.device, max_batch=min(args.batch, max(len(positions), 1)))
    evals = state.evaluate_once(positions)  # chunked internally; relative to the side to move
    if args.pov == "white":
        evals = np.where(positions["stm_ep"] & 0x80, -evals, evals)  # bit 7 = black to move
    positions["eval"] = np.clip(evals, -32768, 32767).astype(np.int16)
    positions.tofile(args.dst)
    print(f"wrote {len(positions)} records to {args.dst} (net '{net.name}')")


if __name__ == "__main__":
    main()
