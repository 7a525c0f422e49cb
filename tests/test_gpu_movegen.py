"""Device-side legal move generation + make-move (spx_movegen) against the host chess core, which is itself pinned by
perft (tests/test_host_logic.py) and by the reference's observer deltas: same legal move sets, byte-identical child
records, same check flags - on random playouts (standard + DFRC), hand-picked castling / en-passant / promotion / check
positions, and through the published perft node counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE_FENS = [
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",      # kiwipete: both castlings, pins
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R b KQkq - 0 1",
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",                                  # ep pin along the rank
    "8/8/8/8/k2Pp2Q/8/8/3K4 b - d3 0 1",                                          # ep capture would expose the king
    "rnbqkb1r/ppp1pppp/5n2/3pP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3",             # plain ep
    "4k3/8/8/8/3pP3/8/8/4K3 b - e3 0 1",
    "r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", "r3k2r/8/8/8/8/8/8/R3K2R b KQkq - 0 1",
    "r3k2r/8/8/8/8/8/6n1/R3K2R w KQkq - 0 1",                                     # knight attacks f1/e... castling through check
    "4k3/8/8/8/8/8/8/R3K2R w KQ - 0 1", "r3k2r/8/8/8/8/5b2/8/R3K2R w KQkq - 0 1",
    "n1n5/PPPk4/8/8/8/8/4Kppp/5N1N b - - 0 1", "n1n5/PPPk4/8/8/8/8/4Kppp/5N1N w - - 0 1",  # promotions incl. captures
    "4k3/8/8/8/8/8/8/4K2q w - - 0 1", "7k/5Q2/6K1/8/8/8/8/8 b - - 0 1",            # check / stalemate
    "R6k/6pp/8/8/8/8/8/4K3 b - - 0 1",                                            # back-rank mate
    "4k3/8/8/8/8/2n5/3b4/4K3 w - - 0 1", "4k3/4r3/8/8/8/2n5/8/4K3 w - - 0 1",      # double check
    "bqnb1rkr/pp3ppp/3ppn2/2p5/5P2/P2P4/NPP1P1PP/BQ1BNRKR w HFhf - 2 9",          # Chess960 castling rights
    "1rk3r1/8/8/8/8/8/8/1RK3R1 w GBgb - 0 1", "rk5r/8/8/8/8/8/8/RK5R w HAha - 0 1", "r5kr/8/8/8/8/8/8/R5KR b HAha - 0 1",
    "2rkr3/8/8/8/8/8/8/2RKR3 w ECec - 0 1", "5rkr/8/8/8/8/8/8/5RKR w HFhf - 0 1",
    "rnbq1bnr/ppppkppp/8/4p3/4P3/8/PPPPKPPP/RNBQ1BNR w - - 2 3", "8/8/8/8/8/8/8/K6k w - - 0 1",
]


def host_children(sp, rec):
    moves, children, chk = sp.legal_moves(rec)
    rows = sorted((int(m), c.tobytes()) for m, c in zip(moves, children))
    return rows, chk


def compare(sp, st, positions):
    out = st.movegen(positions)
    assert int(out["count"].sum()) == len(out["children"])
    for i in range(len(positions)):
        want, chk = host_children(sp, positions[i])
        lo, n = int(out["first"][i]), int(out["count"][i])
        got = sorted((int(m), c.tobytes()) for m, c in zip(out["moves"][lo:lo + n], out["children"][lo:lo + n]))
        fen = sp.position_to_fen(positions[i])
        assert n == len(want), (fen, n, len(want))
        assert got == want, fen
        assert bool(out["in_check"][i]) == chk, fen
        assert np.all(out["parents"][lo:lo + n] == i)
    return out


@pytest.fixture(scope="module")
def st(sp, net_blob):
    s = sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=4096)
    yield s
    s.close()


def test_edge_positions_match_host_core(sp, st):
    compare(sp, st, sp.positions_from_fens(EDGE_FENS))


def test_random_playouts_match_host_core(sp, st):
    pos = np.concatenate([
        sp.random_positions(2500, seed=1, min_ply=0, max_ply=160, dfrc_every=2),
        sp.random_positions(1500, seed=2, min_ply=0, max_ply=14, dfrc_every=1),   # castling rights still alive
    ])
    out = compare(sp, st, pos)
    kinds = out["moves"] >> 14
    assert (kinds == 1).any() and (kinds == 2).any() and (kinds == 3).any()       # ep, castling, promotions all occur
    # parent_values are passed through (e.g. accumulator slots)
    pv = np.arange(len(pos), dtype=np.uint32)[::-1].copy()
    out2 = st.movegen(pos, parent_values=pv)
    for i in (0, 17, len(pos) - 1):
        lo, n = int(out2["first"][i]), int(out2["count"][i])
        assert np.all(out2["parents"][lo:lo + n] == pv[i])


@pytest.mark.parametrize("fen,depth,nodes", [
    ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", 4, 197281),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", 3, 97862),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", 4, 43238),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", 3, 9467),
    ("bqnb1rkr/pp3ppp/3ppn2/2p5/5P2/P2P4/NPP1P1PP/BQ1BNRKR w HFhf - 2 9", 3, 12189),
])
def test_device_perft(sp, st, fen, depth, nodes):
    """Published perft counts (chessprogramming.org positions 1-4; the Chess960 one from the FRC perft suite),
    walked level by level entirely with the device generator."""
    level = sp.positions_from_fens([fen])
    for d in range(depth):
        out = st.movegen(level, capacity=len(level) * 80 + 256)
        if d == depth - 1:
            assert int(out["count"].sum()) == nodes
        level = out["children"]


def test_capacity_overflow_is_reported(sp, st):
    from stormphrax_amd import _lib

    pos = sp.random_positions(64, seed=3)
    with pytest.raises(_lib.SpxError):
        st.movegen(pos, capacity=100)


def test_garbage_records_do_not_fault(sp, st):
    """Random bytes instead of records (no kings, 64 occupied squares, codes 7 / 15): the generator and the incremental
    kernels stay inside their buffers - results for such records are unspecified, results for the valid ones around them
    are not disturbed (hot calls never validate per position, like the reference, which only asserts)."""
    good = sp.random_positions(512, seed=8, min_ply=0, max_ply=100, dfrc_every=3)
    bad = good.copy()
    rng = np.random.default_rng(6)
    raw = bad.view(np.uint8).reshape(-1, 32)
    for i in range(0, 512, 4):
        raw[i] = rng.integers(0, 256, 32, dtype=np.uint8)
    raw[8] = 0xFF
    raw[12] = 0
    want = st.movegen(good, capacity=512 * 256)
    try:
        got = st.movegen(bad, capacity=512 * 256)
    except Exception:  # a capacity overflow from nonsense positions is a legitimate answer
        got = None
    if got is not None:
        for i in range(512):
            if i % 4 == 0:
                continue
            lo, n = int(got["first"][i]), int(got["count"][i])
            wlo, wn = int(want["first"][i]), int(want["count"][i])
            assert n == wn
            a = sorted((int(m), c.tobytes()) for m, c in zip(got["moves"][lo:lo + n], got["children"][lo:lo + n]))
            b = sorted((int(m), c.tobytes()) for m, c in zip(want["moves"][wlo:wlo + wn], want["children"][wlo:wlo + wn]))
            assert a == b
    # incremental kernels: garbage parents / children among valid pairs
    st.reserve_slots(2048)
    slots = np.arange(512, dtype=np.uint32)
    st.reset(bad, slots)                                  # refresh from garbage records
    nxt, moved = sp.random_successors(good, seed=2)
    mixed = nxt.copy()
    mixed[1::4] = bad[0::4]                               # garbage children for valid parents
    out = st.update_evaluate(slots, slots + 1024, mixed)  # parents 0,4,8.. hold garbage records themselves
    full = st.evaluate_once(nxt)
    ok = np.ones(512, dtype=bool)
    ok[0::4] = False
    ok[1::4] = False
    assert np.array_equal(out[ok], full[ok])
    assert np.array_equal(st.evaluate_once(good), st.evaluate_once(good.copy()))  # the context is still healthy


def test_viriformat_expansion_on_the_device_matches_the_host_replay(sp, st):
    """spx_viri_expand_gpu (one thread per game replays the moves on the packed records) against the host expander,
    which validates every move with the legal-move generator: byte-identical records over random games with castling
    (standard and Chess960), en passant and promotions; self-play output of the driver goes through the same check."""
    import time

    blob = b"".join(sp.viri_random_game(3000 + seed, plies=180, dfrc=(seed % 3 == 0)) for seed in range(300))
    t0 = time.perf_counter()
    want, games = sp.viri_expand(blob)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    got, games_gpu, bad = st.viri_expand(blob)
    t_gpu = time.perf_counter() - t0
    assert games_gpu == games == 300 and bad == 0 and len(got) == len(want) > 20000
    assert got.tobytes() == want.tobytes()
    # the marlinformat filter (in check / noisy move) computed on the device equals the host's
    _, _, keep_host = sp.viri_expand(blob, with_filter=True)
    _, _, _, keep_gpu = st.viri_expand(blob, with_filter=True)
    assert np.array_equal(keep_gpu, keep_host) and 0.3 < keep_host.mean() < 0.95
    # ... and so does the game-level rule (the move that ends a game in a draw is filtered whatever it is, datagen.cpp:264-268) on
    # games that end in repetitions, bare material and on the 50-move clock (tests/golden/drawn_games.txt, cut where they are drawn)
    import os

    from test_host_logic import _viri_stream

    drawn_blob, n_drawn = b"", 0
    for ln in open(os.path.join(os.path.dirname(__file__), "golden", "drawn_games.txt")):
        if ln.startswith("#"):
            continue
        start, _, line = ln.rstrip("\n").split(" | ")
        items = [it.rsplit(":", 1) for it in line.split()]
        cut = next((k for k, (_, w) in enumerate(items) if int(w)), len(items) - 1)
        drawn_blob += _viri_stream(sp, start, [u for u, _ in items[:cut + 1]])
        n_drawn += 1
    _, games_host, keep_host = sp.viri_expand(drawn_blob, with_filter=True)
    _, games_dev, bad, keep_gpu = st.viri_expand(drawn_blob, with_filter=True)
    assert games_host == games_dev == n_drawn >= 70 and bad == 0 and np.array_equal(keep_gpu, keep_host)
    kinds = set()
    off = 0
    while off < len(blob):   # every move type occurs in the input
        off += 32
        while blob[off:off + 4] != b"\x00\x00\x00\x00":
            kinds.add(blob[off + 1] >> 6)
            off += 4
        off += 4
    assert kinds == {0, 1, 2, 3}
    print(f"viri expand: host {len(want) / t_host:.3e} positions/s, device path {len(got) / t_gpu:.3e} (incl. copies)")
    # a corrupted move (from-square without a piece of the side to move) is counted, not followed
    broken = bytearray(blob)
    broken[32] = 27 | (broken[32] & 0xC0)  # first move of the first game now starts on d4 - empty in every start position
    broken[33] &= 0xF0
    _, _, bad = st.viri_expand(bytes(broken))
    assert bad == 1


def test_random_playouts_on_the_device_are_legal_positions(sp, oracle, net_blob):
    """spx_random_positions_gpu: seeded, reproducible batches of random-playout positions written straight to device
    memory (page-locked host memory here, which the device addresses). Every record must be a position the host chess
    core accepts and reproduces (FEN round trip), evaluate like the CPU oracle, and contain Chess960 starts."""
    import ctypes

    from stormphrax_amd import _lib

    lib = _lib.load()
    st = sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=8192)
    n = 6000
    bufs = [lib.spx_host_alloc(n * 32) for _ in range(3)]
    try:
        assert all(bufs)
        views = [np.ctypeslib.as_array((ctypes.c_uint8 * (n * 32)).from_address(b)) for b in bufs]
        st.random_positions_device(bufs[0], n, seed=5, min_ply=6, max_ply=90, dfrc_every=3)
        st.random_positions_device(bufs[1], n, seed=5, min_ply=6, max_ply=90, dfrc_every=3)
        st.random_positions_device(bufs[2], n, seed=6, min_ply=6, max_ply=90, dfrc_every=3)
        assert np.array_equal(views[0], views[1]) and not np.array_equal(views[0], views[2])
        pos = views[0].copy().view(sp.PACKED_DTYPE)
        fens = [sp.position_to_fen(p) for p in pos]
        again = sp.positions_from_fens(fens)
        assert again.tobytes() == pos.tobytes(), "a generated record is not what the host core packs for its own FEN"
        assert len(set(fens)) > 0.95 * n
        mail, stm = sp.positions_to_mailboxes(pos[:2048])
        oracle.use(net_blob("tame"), "tame")
        assert np.array_equal(st.evaluate_once(pos[:2048]), oracle.eval_mailboxes(mail, stm))
        plies = np.array([2 * (int(f.split()[-1]) - 1) + (f.split()[1] == "b") for f in fens])
        assert plies.min() >= 6 and plies.max() <= 90 and plies.std() > 15
    finally:
        for b in bufs:
            lib.spx_host_free(b)
        st.close()
