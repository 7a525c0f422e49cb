"""CPU-side model of the column-sliced gather's wave loads (round 5, VERDICT r4 items 1 + 3): how many vector-memory
instructions (the binding unit: every 16-byte-per-lane wave load holds the CU's texture path for ~17 cycles whatever its exec
mask or width - tools/probes/tcp_mask_probe.hip) and how many LDS reads does a design need for the bench batch?

Feature lists come from the ORACLE (test infrastructure, runs on the host), positions from the bench's own generator
(seed 20260927, plies 8-120, DFRC every 4th). Designs:
  base        round 4: sections [high planes][piece-square (LDS slab)][threat rows (global)], sort key = total quartets
  hot N       + the N most popular threat / pawn-pair rows (measured on a calibration batch of a DIFFERENT seed) in LDS:
              sections [slab][hot (LDS)][cold (global)]; key variants: total quartets / cold quartets major
  pairs       ... and every perspective pair of a group walks its own number of quartets (no group-wide padding)
  octets      ... and every MFMA adds 8 rows of ONE perspective (two half lists side by side), no padding to neighbours at all
Output: global wave loads and LDS wave reads per POSITION (all 8 slices), i.e. x 65 536 = per launch of the bench batch.

    python tools/sim_gather_steps.py [--positions 8192] [--preset tame]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def feature_lists(sp, oracle, pos):
    mail, _ = sp.positions_to_mailboxes(pos)
    out = []
    for i in range(len(pos)):
        for c in (0, 1):
            psq, thr = oracle.features(mail[i], c)
            out.append((psq.copy(), thr.copy()))
    return out


def king_bucket_of(psq_rows):
    return int(psq_rows[0]) // 704 if len(psq_rows) else 0


def q4(n):
    return (n + 3) // 4


def simulate(lists, hot_rank, n_hot, key_mode, form, wide_rows=None, hi_slot=None):
    """-> (global wave loads, LDS wave reads) per position, summed over the 8 slices."""
    persp = []
    for psq, thr in lists:
        b = king_bucket_of(psq)
        hot = int(np.count_nonzero(hot_rank[thr] < n_hot)) if n_hot else 0
        cold = len(thr) - hot
        hi_cold = hi_hot = 0
        if wide_rows is not None:
            w = psq[wide_rows[psq]]
            if hi_slot is not None:
                hi_hot = int(np.count_nonzero(hi_slot[w]))
            hi_cold = len(w) - hi_hot
        persp.append((b, len(psq), hot, cold, hi_cold, hi_hot))
    persp = np.array(persp, dtype=np.int64)
    b, npsq, hot, cold, hic, hih = persp.T
    lds_q = q4(npsq + hot + hih) if form.startswith("stages") else q4(npsq) + q4(hot) + q4(hih)
    glob_q = q4(cold) + q4(hic)
    if key_mode == "total":
        key = b * 100000 + np.minimum(lds_q + glob_q, 79)
    elif key_mode == "cold":
        key = b * 100000 + np.minimum(glob_q, 39) * 100 + np.minimum(lds_q, 99) // 4  # (39 x 25 bins... a model, not the kernel's key)
    elif key_mode == "cold_exact":
        key = b * 100000 + glob_q * 100 + lds_q
    elif key_mode.startswith("bins"):  # bins<C>x<L>x<S>: cold quartets clamped to C values, LDS quartets >> S clamped to L values
        C, L, S = map(int, key_mode[4:].split("x"))
        key = b * 100000 + np.minimum(glob_q, C - 1) * 100 + np.minimum(lds_q >> S, L - 1)
    order = np.argsort(key, kind="stable")
    g_loads = l_reads = 0
    # groups of 8 inside a bucket
    for bucket in range(16):
        idx = order[b[order] == bucket]
        for s in range(0, len(idx), 8):
            grp = idx[s:s + 8]
            secs_l = [q4(npsq[grp]), q4(hot[grp]), q4(hih[grp])]
            secs_g = [q4(cold[grp]), q4(hic[grp])]
            if form == "group":      # 4 wave loads per step, steps = the longest list's quartets, per section
                l_reads += 4 * sum(int(x.max()) for x in secs_l)
                g_loads += 4 * sum(int(x.max()) for x in secs_g)
            elif form == "stages":   # round 5: the LDS section merged (piece-square + hot rows), stages of 8 steps walked in pairs
                def even_stages(q):
                    full, rest = divmod(int(q), 8)
                    return 8 * full + ((rest + 1) & ~1)
                lq = q4(npsq[grp] + hot[grp] + hih[grp])
                l_reads += 4 * even_stages(lq.max())
                g_loads += 4 * sum(even_stages(x.max()) for x in secs_g)
            elif form == "stages_exact":  # ... without the padding to pairs
                lq = q4(npsq[grp] + hot[grp] + hih[grp])
                l_reads += 4 * int(lq.max())
                g_loads += 4 * sum(int(x.max()) for x in secs_g)
            elif form == "pairs":    # every perspective pair walks its own quartets
                pad = np.zeros(8, dtype=np.int64)
                for x in secs_l:
                    pad[:len(grp)] = x
                    pad[len(grp):] = 0
                    l_reads += int(np.maximum(pad[0::2], pad[1::2]).sum())
                for x in secs_g:
                    pad[:len(grp)] = x
                    pad[len(grp):] = 0
                    g_loads += int(np.maximum(pad[0::2], pad[1::2]).sum())
            elif form == "octets":   # one wave load = 8 rows of ONE perspective
                l_reads += int(((npsq[grp] + 7) // 8 + (hot[grp] + 7) // 8 + (hih[grp] + 7) // 8).sum())
                g_loads += int(((cold[grp] + 7) // 8 + (hic[grp] + 7) // 8).sum())
    n_pos = len(lists) / 2
    return 8 * g_loads / n_pos, 8 * l_reads / n_pos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=8192)
    ap.add_argument("--preset", default="tame")
    args = ap.parse_args()
    import stormphrax_amd as sp
    from conftest import Oracle

    oracle = Oracle()
    blob = sp.synthetic_net_bytes(args.preset)
    oracle.use(blob, args.preset)
    calib = feature_lists(sp, oracle, sp.random_positions(args.positions, seed=4711, min_ply=8, max_ply=120, dfrc_every=4))
    lists = feature_lists(sp, oracle, sp.random_positions(args.positions, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4))
    counts = np.zeros(64368, dtype=np.int64)
    for _, thr in calib:
        np.add.at(counts, thr, 1)
    rank_order = np.argsort(-counts, kind="stable")
    hot_rank = np.empty(64368, dtype=np.int64)
    hot_rank[rank_order] = np.arange(64368)
    fetched = sum(len(t) for _, t in lists)
    print(f"{args.positions} positions: threat + pawn-pair rows per position {fetched / args.positions:.1f}, piece-square rows "
          f"{sum(len(p) for p, _ in lists) / args.positions:.1f}")
    for n in (64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 1024):
        hit = sum(int(np.count_nonzero(hot_rank[t] < n)) for _, t in lists)
        print(f"  hot {n:5d}: {100.0 * hit / fetched:5.1f} % of the threat / pawn-pair fetches")
    wide = None
    if args.preset != "tame":
        psq = blob[64:64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024)
        wide = (psq.min(axis=1) < -128) | (psq.max(axis=1) > 127)
        print(f"  piece-square rows with weights outside i8: {int(wide.sum())} of 11264")
    print(f"{'design':58s} {'global loads / pos':>20s} {'LDS reads / pos':>18s}   (x 65 536 = per launch)")
    base = simulate(lists, hot_rank, 0, "total", "group", wide)
    print(f"{'base (round 4)':58s} {base[0]:20.1f} {base[1]:18.1f}")
    for n in (256, 320):
        for key_mode in ("cold_exact", "bins16x5x2"):
            for form in ("group", "stages_exact", "stages"):
                g, l = simulate(lists, hot_rank, n, key_mode, form, wide)
                print(f"{'hot %d, key %s, %s' % (n, key_mode, form):58s} {g:20.1f} {l:18.1f}   global x {g / base[0]:.3f}")


if __name__ == "__main__":
    main()
