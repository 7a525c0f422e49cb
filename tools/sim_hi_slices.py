"""CPU-side model (round 6) of what dropping the all-zero slices of high-byte planes buys the column-sliced gather on a net with wide
piece-square rows: global wave loads and LDS wave reads per POSITION (all 8 slices) for
  shared        round 5: one high-byte section for all XCDs (every wide row's plane is fetched by every XCD)
  slice         round 6: XCD x walks only the planes that are not all zero in slice x (spx_ftx_gather_kernel compacts the stage)
  slice_nonear / slice_dense   ... if rows with <= 8 / <= 64 weights outside i8 brought no plane at all (what a sparse-remainder
                scheme could reach at best)
and for LDS-resident high planes beside fewer hot threat rows. Feature lists come from the ORACLE (test infrastructure).

    python tools/sim_hi_slices.py [positions] [preset]        # measured on the GPU: 108 -> 91 row loads per position (realistic)
"""
import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import stormphrax_amd as sp
from conftest import Oracle
from sim_gather_steps import feature_lists, q4

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
preset = sys.argv[2] if len(sys.argv) > 2 else "realistic"
oracle = Oracle()
blob = sp.synthetic_net_bytes(preset)
oracle.use(blob, preset)
calib = feature_lists(sp, oracle, sp.random_positions(N, seed=4711, min_ply=8, max_ply=120, dfrc_every=4))
lists = feature_lists(sp, oracle, sp.random_positions(N, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4))
counts = np.zeros(64368, dtype=np.int64)
for _, thr in calib:
    np.add.at(counts, thr, 1)
rank_order = np.argsort(-counts, kind="stable")
hot_rank = np.empty(64368, dtype=np.int64); hot_rank[rank_order] = np.arange(64368)
psqW = blob[64:64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024).astype(np.int32)
out = (psqW < -128) | (psqW > 127)
nout = out.sum(axis=1)
wide = nout > 0
# slice of column c: (c & 511) >> 6
slice_of = (np.arange(1024) & 511) >> 6
mask = np.zeros((11264, 8), dtype=bool)
for x in range(8):
    mask[:, x] = out[:, slice_of == x].any(axis=1)
print("rows wide", wide.sum(), "by outliers: <=8:", ((nout > 0) & (nout <= 8)).sum(), "9..64:", ((nout > 8) & (nout <= 64)).sum(), ">64:", (nout > 64).sum())
print("mean nonzero slices per wide row", mask[wide].sum(axis=1).mean())
# psq hi row popularity per bucket
hicount = np.zeros(11264, dtype=np.int64)
for psq, _ in calib:
    np.add.at(hicount, psq[wide[psq]], 1)

def even_stages(q):
    full, rest = divmod(int(q), 8)
    return 8 * full + ((rest + 1) & ~1)

def run(n_hot, mode, hi_lds_rows=0, coldShift=0):
    # mode: 'shared' (current), 'slice' (per-slice compaction), 'slice_nonear' (near rows' hi handled elsewhere: <=8 outliers), 'slice_dense' (only >64 outliers keep planes)
    P = []
    # LDS-resident hi rows per bucket: top hi_lds_rows by popularity within bucket
    hi_res = np.zeros(11264, dtype=bool)
    if hi_lds_rows:
        for b in range(16):
            rows = np.arange(704 * b, 704 * (b + 1))
            top = rows[np.argsort(-hicount[rows], kind="stable")[:hi_lds_rows]]
            hi_res[top[hicount[top] > 0]] = True
    for psq, thr in lists:
        b = int(psq[0]) // 704
        hot = int(np.count_nonzero(hot_rank[thr] < n_hot))
        cold = len(thr) - hot
        w = psq[wide[psq]]
        if mode == 'slice_nonear': w = w[nout[w] > 8]
        if mode == 'slice_dense': w = w[nout[w] > 64]
        wl = w[hi_res[w]]; wg = w[~hi_res[w]]
        if mode == 'shared':
            hx_g = np.full(8, len(wg)); hx_l = np.full(8, len(wl))
        else:
            hx_g = mask[wg].sum(axis=0); hx_l = mask[wl].sum(axis=0)
        P.append((b, len(psq), hot, cold, len(w)) + tuple(hx_g) + tuple(hx_l))
    P = np.array(P, dtype=np.int64)
    b, npsq, hot, cold, nhi = P[:, 0], P[:, 1], P[:, 2], P[:, 3], P[:, 4]
    hxg = P[:, 5:13]; hxl = P[:, 13:21]
    lds_q = q4(npsq + hot); glob_q = q4(cold) + q4(nhi)
    # kernel's key: bucket*80 + min(glob_q>>s,15)*5 + min(lds_q>>2,4)
    key = b * 100000 + np.minimum(glob_q >> coldShift, 15) * 100 + np.minimum(lds_q >> 2, 4)
    order = np.argsort(key, kind="stable")
    g = l = 0
    for bucket in range(16):
        idx = order[b[order] == bucket]
        for s in range(0, len(idx), 8):
            grp = idx[s:s + 8]
            l += 8 * 4 * even_stages(q4(npsq[grp] + hot[grp]).max())
            g += 8 * 4 * even_stages(q4(cold[grp]).max())
            for x in range(8):
                g += 4 * even_stages(q4(hxg[grp, x]).max())
                l += 4 * even_stages(q4(hxl[grp, x]).max())
    npos = len(lists) / 2
    return g / npos, l / npos

for n_hot in (256,):
    for mode in ('shared', 'slice', 'slice_nonear', 'slice_dense'):
        for cs in (0, 1):
            g, l = run(n_hot, mode, 0, cs)
            print(f"hot {n_hot} {mode:14s} coldShift {cs}: global loads/pos {g:7.1f}  LDS reads/pos {l:7.1f}  cost(17.5g+8.8l) {17.5*g+8.8*l:8.0f}")
for n_hot, hl in ((192, 64), (160, 96), (128, 128), (256, 64)):
    for mode in ('shared', 'slice'):
        g, l = run(n_hot, mode, hl, 1)
        print(f"hot {n_hot} hiLDS {hl} {mode:8s}: global loads/pos {g:7.1f}  LDS reads/pos {l:7.1f}  cost {17.5*g+8.8*l:8.0f}")
