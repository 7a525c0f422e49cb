"""before carrying the hot-row set of the full-refresh gather over to the UPDATE kernel, histogram the rows a
MOVE touches. CPU only (the host emulation of the update kernel's delta derivation, spx_debug_delta): random legal positions of the
bench's generator, one random legal move each (what `bench.py --mode incremental` and a search's make-move do), both perspectives.

    python tools/delta_row_popularity.py [--positions 20000] > profiles/r05_delta_row_popularity.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=20000)
    args = ap.parse_args()
    import stormphrax_amd as sp

    rng = np.random.default_rng(5)
    pos = sp.random_positions(args.positions, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
    thr_delta = np.zeros(64368, dtype=np.int64)
    psq_delta = np.zeros(11264, dtype=np.int64)
    thr_full = np.zeros(64368, dtype=np.int64)
    moves = refreshes = 0
    per_move = []
    for i in range(len(pos)):
        words, kids, _ = sp.legal_moves(pos[i])
        if len(words) == 0:
            continue
        child = kids[int(rng.integers(len(words)))]
        moves += 1
        n_rows = 0
        for c in (0, 1):
            d = sp.debug_delta(pos[i], child, c)
            if d["refresh"]:
                refreshes += 1
                continue
            for key in ("thr_sub", "thr_add"):
                np.add.at(thr_delta, d[key], 1)
                n_rows += len(d[key])
            for key in ("psq_sub", "psq_add"):
                np.add.at(psq_delta, d[key], 1)
                n_rows += len(d[key])
            _, thr = sp.debug_features(pos[i], c)
            np.add.at(thr_full, thr, 1)
        per_move.append(n_rows)
    print(f"{moves} random legal moves from the bench generator's positions (seed 20260927, plies 8-120, every 4th game DFRC), both "
          f"perspectives; {refreshes} perspective refreshes (king bucket / mirror changes) left out")
    print(f"delta rows per move (both perspectives): mean {np.mean(per_move):.1f}, median {np.median(per_move):.0f}, p95 {np.percentile(per_move, 95):.0f}")
    total_thr, total_full = thr_delta.sum(), thr_full.sum()
    print(f"threat / pawn-pair delta rows: {total_thr} fetches over {int(np.count_nonzero(thr_delta))} distinct rows of 64 368; "
          f"piece-square delta rows: {psq_delta.sum()} over {int(np.count_nonzero(psq_delta))} of 11 264")
    by_delta = np.sort(thr_delta)[::-1]
    by_full = np.sort(thr_full)[::-1]
    # a hot set chosen from FULL-REFRESH popularity (what the gather's calibration would pick) applied to the delta fetches
    order_full = np.argsort(-thr_full, kind="stable")
    print("share of the threat / pawn-pair fetches served by the N most popular rows:")
    print("      N   full refresh (own ranking)   move deltas (own ranking)   move deltas (full-refresh ranking)")
    for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        print(f"  {n:5d}   {100 * by_full[:n].sum() / total_full:22.1f} %   {100 * by_delta[:n].sum() / total_thr:22.1f} %   "
              f"{100 * thr_delta[order_full[:n]].sum() / total_thr:30.1f} %")
    print("the 12 most popular delta rows (row id, fetches, share):",
          ", ".join(f"{int(r)}: {int(thr_delta[r])} ({100 * thr_delta[r] / total_thr:.2f} %)" for r in np.argsort(-thr_delta)[:12]))


if __name__ == "__main__":
    main()
