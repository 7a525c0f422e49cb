"""GPU parity: the HIP path (through the C ABI) must equal the CPU oracle bit for bit.

Reference behaviour under test: NnueState::evaluateOnce (src/eval/nnue_state.cpp:612-634)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STARTPOS = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"


@pytest.fixture(scope="module")
def states(sp, net_blob):
    cache = {}

    def get(preset):
        if preset not in cache:
            cache[preset] = sp.NnueState(sp.Network(net_blob(preset)), device=0, max_batch=1 << 16)
        return cache[preset]

    yield get
    for s in cache.values():
        s.close()


PATHS = ["one_kernel", "sliced", "sliced_pipelined"]


@pytest.fixture(scope="module")
def path_states(sp, net_blob):
    """One context per (preset, full-refresh path): `one_kernel` = spx_ft_kernel (SPX_CTX_ONE_KERNEL_FT), `sliced` = the
    column-sliced pipeline of spx_ftx.hip - the default path of big batches - forced onto every batch of >= 1 024 positions,
    `sliced_pipelined` = the same context driven through spx_eval_full_device_async (the reference's goldens
    must travel the default path in what the driver re-runs)."""
    cache = {}

    def get(preset, path):
        key = (preset, "one_kernel" if path == "one_kernel" else "sliced")
        if key not in cache:
            if path == "one_kernel":
                cache[key] = sp.NnueState(sp.Network(net_blob(preset)), device=0, max_batch=1 << 16, sliced_ft=False)
            else:
                # (mlp_share_max = 0: the MLP in its big-batch form too - one wave per tile, the L2 weights streamed)
                cache[key] = _state_with_options(sp, net_blob(preset), {"ftx_min": 1024, "tiny_batch_max": 0, "mlp_share_max": 0},
                                                 max_batch=1 << 16)
                assert cache[key].takes_sliced_pipeline(1024) and cache[key].takes_sliced_pipeline(1024, pipelined=True)
        return cache[key]

    yield get
    for s in cache.values():
        s.close()


def _evaluate_through(sp, st, pos, path):
    """evaluate_once over `path`; the pipelined entry point reads / writes page-locked host memory the device addresses."""
    if path != "sliced_pipelined":
        return st.evaluate_once(pos)
    import ctypes

    from stormphrax_amd import _lib

    lib = _lib.load()
    n = len(pos)
    pin, pout = lib.spx_host_alloc(n * 32), lib.spx_host_alloc(n * 4)
    assert pin and pout
    try:
        np.ctypeslib.as_array((ctypes.c_uint8 * (n * 32)).from_address(pin))[:] = pos.view(np.uint8).reshape(-1)
        out = np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(pout))
        out[:] = -1
        # two calls in flight on the two lanes (the second one's preparation runs beside the first one's gather)
        half = n // 2
        assert st.evaluate_once_device_async(pin, half, pout)
        assert st.evaluate_once_device_async(pin + 32 * half, n - half, pout + 4 * half)
        st.synchronize()
        return out.copy()
    finally:
        lib.spx_host_free(pin)
        lib.spx_host_free(pout)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("preset", ["tame", "wild", "extreme", "realistic"])
def test_random_positions_bit_exact(sp, oracle, net_blob, path_states, preset, path):
    pos = sp.random_positions(4096, seed=101, min_ply=0, max_ply=160, dfrc_every=3)
    mail, stm = sp.positions_to_mailboxes(pos)
    oracle.use(net_blob(preset), preset)
    want = oracle.eval_mailboxes(mail, stm)
    got = _evaluate_through(sp, path_states(preset, path), pos, path)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first: {sp.position_to_fen(pos[bad[0]])} got {got[bad[0]]} want {want[bad[0]]}"
    assert len(set(want.tolist())) > (500 if preset == "realistic" else 1000)  # the batch is not degenerate (the heavy-tailed net spreads evals less)


@pytest.mark.parametrize("n", [16384, 20011, 65536])
def test_the_pipeline_sorts_the_mlps_order_itself(sp, oracle, net_blob, n):
    """Round 6: a one-pass batch of the column-sliced pipeline gets the MLP's output-bucket order (output.h:44-55) from the
    pipeline's own counting sort - the extraction notes the bucket, the rank / plan / scatter kernels order the positions - instead of
    two spx_sort_* launches (option ftx_fold_sort = 0). Same scores either way, and the oracle's; every output bucket occurs."""
    pos = sp.random_positions(n, seed=4242, min_ply=0, max_ply=200, dfrc_every=3)
    with _state_with_options(sp, net_blob("wild"), {"ftx_fold_sort": 1}, max_batch=n) as folded, \
            _state_with_options(sp, net_blob("wild"), {"ftx_fold_sort": 0}, max_batch=n) as plain:
        assert folded.takes_sliced_pipeline(n)
        got = folded.evaluate_once(pos)
        assert np.array_equal(got, plain.evaluate_once(pos))
        assert np.array_equal(folded.evaluate_once(pos[: n - 77]), got[: n - 77])  # (the counters are back at zero for the next batch)
    buckets = {(bin(int(o)).count("1") - 2) // 4 for o in pos["occupancy"][:4000]}
    assert len(buckets) >= 7
    mail, stm = sp.positions_to_mailboxes(pos[:2500])
    oracle.use(net_blob("wild"), "wild")
    assert np.array_equal(got[:2500], oracle.eval_mailboxes(mail, stm))


def test_ft_activations_bit_exact(sp, oracle, net_blob, states):
    """Localises failures: the feature-transformer kernel's u8 activations vs activateFt (multilayer.h:92-152)."""
    oracle.use(net_blob("extreme"), "extreme")
    pos = sp.random_positions(200, seed=55)
    mail, stm = sp.positions_to_mailboxes(pos)
    st = states("extreme")
    st.evaluate_once(pos)
    got = st.debug_ft(len(pos))
    for i in range(len(pos)):
        want = oracle.ft(mail[i], stm[i])
        assert np.array_equal(got[i], want), f"position {i}: {sp.position_to_fen(pos[i])}"


def test_ragged_and_tiny_batches(sp, oracle, net_blob, states):
    oracle.use(net_blob("tame"), "tame")
    st = states("tame")
    pos = sp.random_positions(300, seed=7)
    mail, stm = sp.positions_to_mailboxes(pos)
    want = oracle.eval_mailboxes(mail, stm)
    for n in (1, 2, 3, 63, 64, 65, 127, 129, 300):
        got = st.evaluate_once(pos[:n])
        assert np.array_equal(got, want[:n]), n
    assert st.evaluate_once(pos[:0]).shape == (0,)


def test_startpos_and_bare_kings(sp, oracle, net_blob, states):
    oracle.use(net_blob("tame"), "tame")
    fens = [STARTPOS, "8/8/4k3/8/8/3K4/8/8 w - - 0 1", "8/8/4k3/8/8/3K4/8/8 b - - 0 1",
            "k7/8/8/8/8/8/8/7K w - - 0 1", "7k/8/8/8/8/8/8/K7 b - - 0 1",
            # 32 pieces with many mutual threats and the maximum number of pawns
            "r1bqkb1r/pppppppp/2n2n2/8/8/2N2N2/PPPPPPPP/R1BQKB1R w KQkq - 4 3"]
    pos = sp.positions_from_fens(fens)
    got = states("tame").evaluate_once(pos)
    want = np.array([oracle.eval_fen(f) for f in fens], dtype=np.int32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("preset", ["tame", "wild", "extreme", "realistic"])
def test_reference_golden_vectors(sp, net_blob, path_states, preset, path):
    """GPU vs the COMPILED REFERENCE directly (tests/golden/evals.jsonl), without going through the oracle - through the
    one-kernel path, the column-sliced pipeline (spx_ftx_gather_kernel: the default path of big batches) and its pipelined
    entry point."""
    import json
    import os

    golden = os.path.join(os.path.dirname(__file__), "golden", "evals.jsonl")
    recs = [json.loads(line) for line in open(golden)]
    pos = sp.positions_from_fens([r["fen"] for r in recs])
    got = _evaluate_through(sp, path_states(preset, path), pos, path)
    want = np.array([r[preset] for r in recs], dtype=np.int32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (recs[bad[0]]["fen"], int(got[bad[0]]), int(want[bad[0]]))


@pytest.mark.parametrize("preset", ["tame", "realistic"])
def test_extraction_lists_equal_the_reference_feature_rows(sp, net_blob, path_states, preset):
    """what spx_ftx_extract_kernel lists for a perspective - piece-square rows (LDS slab), hot and cold
    threat / pawn-pair rows, and on the heavy-tailed net the high-byte planes of the wide piece-square rows - decoded back
    to the net's row numbering (spx_debug_ftx_lists) and compared AS MULTISETS with the COMPILED REFERENCE's active features
    (tests/golden/features.jsonl: psq.h:338-365, threats.cpp:170-221, nnue_state.cpp:309-354). No oracle in between."""
    import json
    import os
    from collections import Counter

    recs = [json.loads(line) for line in open(os.path.join(os.path.dirname(__file__), "golden", "features.jsonl"))]
    assert len(recs) >= 300
    fens = [r["fen"] for r in recs]
    pos = sp.positions_from_fens(fens * (1 + 1024 // len(fens)))  # (the sliced pipeline takes batches of >= 1 024 here)
    st = path_states(preset, "sliced")
    st.evaluate_once(pos)
    lists = st.ftx_lists(len(pos))
    blob = net_blob(preset)
    psq_w = blob[64:64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024)
    wide = (psq_w.min(axis=1) < -128) | (psq_w.max(axis=1) > 127)
    assert bool(wide.any()) == (preset == "realistic")
    hot = set(int(r) for r in st.hot_rows())
    from_hot = 0
    for i in range(len(pos)):
        r = recs[i % len(recs)]
        for c in (0, 1):  # (the lists are kept per COLOUR, 0 = black; the head's output slot puts the side to move's half first)
            psq, thr, high = lists[2 * i + c]
            assert Counter(psq.tolist()) == Counter(r["psq"][c]), (r["fen"], c)
            assert Counter(thr.tolist()) == Counter(r["thr"][c]), (r["fen"], c)
            assert Counter(high.tolist()) == Counter(row for row in r["psq"][c] if wide[row]), (r["fen"], c)
            from_hot += sum(1 for row in r["thr"][c] if row in hot)
    assert from_hot > 0  # the lists went through the hot-set split


def test_full_size_batch_properties(sp, oracle, net_blob, states):
    """BASELINE config 2 at full size (65 536 positions): size-independent properties instead of a full oracle pass.
    (1) order independence: a shuffled batch gives the shuffled result; (2) batch-split independence: two halves equal
    the whole; (3) a 2 048-position random sample equals the oracle; (4) colour-flip symmetry is NOT assumed (the net is
    random) but stm dependence is: flipping only the stm bit changes the perspective order, checked against the oracle."""
    st = states("wild")
    oracle.use(net_blob("wild"), "wild")
    pos = sp.random_positions(65536, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
    full = st.evaluate_once(pos)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(pos))
    assert np.array_equal(st.evaluate_once(pos[perm]), full[perm])
    halves = np.concatenate([st.evaluate_once(pos[:30001]), st.evaluate_once(pos[30001:])])
    assert np.array_equal(halves, full)
    idx = rng.choice(len(pos), 2048, replace=False)
    mail, stm = sp.positions_to_mailboxes(pos[idx])
    assert np.array_equal(full[idx], oracle.eval_mailboxes(mail, stm))
    flipped = pos[idx].copy()
    flipped["stm_ep"] ^= 0x80
    assert np.array_equal(st.evaluate_once(flipped), oracle.eval_mailboxes(mail, 1 - stm))
    # checksum of checksums: a stable digest of the whole batch for cross-run comparison
    assert int(full.astype(np.int64).sum()) == int(halves.astype(np.int64).sum())


def test_malformed_records_do_not_fault(sp, states):
    """Garbage in the batch (no kings, 64 occupied squares, invalid nibbles) must neither crash the device nor disturb
    the valid records around it: hot calls do not validate per position (like the reference, which only asserts)."""
    st = states("tame")
    good = sp.random_positions(256, seed=3)
    want = st.evaluate_once(good)
    bad = good.copy()
    rng = np.random.default_rng(5)
    raw = bad.view(np.uint8).reshape(-1, 32)
    for i in range(0, 256, 4):
        raw[i] = rng.integers(0, 256, 32, dtype=np.uint8)
    raw[8] = 0xFF           # all squares occupied, every nibble 0xF
    raw[12] = 0             # empty board
    got = st.evaluate_once(bad)
    keep = np.ones(256, dtype=bool)
    keep[::4] = False
    assert np.array_equal(got[keep], want[keep])
    assert np.array_equal(st.evaluate_once(good), want)  # the context is still healthy


def test_malformed_records_in_a_pipeline_sized_batch(sp, net_blob):
    """The same through the column-sliced pipeline (30 000 records, every fourth one garbage - random bytes, 64 occupied squares with
    every nibble 0xF, an empty board): its extraction caps, sort keys and lists must hold for boards no game produces, the valid
    records around them keep their scores, and a second, clean batch comes out right."""
    n = 30000
    good = sp.random_positions(n, seed=13)
    with sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=n, sliced_ft=False) as plain, \
            sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=n) as st:
        assert st.takes_sliced_pipeline(n)
        want = plain.evaluate_once(good)
        bad = good.copy()
        rng = np.random.default_rng(6)
        raw = bad.view(np.uint8).reshape(-1, 32)
        raw[::4] = rng.integers(0, 256, (len(raw[::4]), 32), dtype=np.uint8)
        raw[8] = 0xFF
        raw[12] = 0
        raw[16, :8] = 0xFF  # 64 pieces, valid nibbles: queens and knights everywhere
        raw[16, 8:24] = 0x49
        got = st.evaluate_once(bad)
        keep = np.ones(n, dtype=bool)
        keep[::4] = False
        assert np.array_equal(got[keep], want[keep])
        assert np.array_equal(st.evaluate_once(good), want)


def test_mixed_compact_and_wide_piece_square_rows(sp, oracle, net_blob, states):
    """Piece-square rows whose weights all fit i8 are served from a 1 KiB u8 copy, the others from the i16 table.
    A net with both kinds (conftest._mixed_rows_net) must still equal the oracle bit for bit; the per-row
    classification is checked through the context's count."""
    from conftest import MIXED_WIDE_ROWS

    st = states("mixed")
    assert st.compact_psq_rows == int((~MIXED_WIDE_ROWS).sum())
    assert states("tame").compact_psq_rows == 11264 and states("extreme").compact_psq_rows < 11264
    pos = sp.random_positions(4096, seed=77, min_ply=0, max_ply=140, dfrc_every=4)
    mail, stm = sp.positions_to_mailboxes(pos)
    oracle.use(net_blob("mixed"), "mixed")
    want = oracle.eval_mailboxes(mail, stm)
    assert np.array_equal(st.evaluate_once(pos), want)
    # accumulators written by the refresh path (same lists) feed the arena path unchanged
    st.reserve_slots(4096)
    st.reset(pos, np.arange(4096, dtype=np.uint32))
    assert np.array_equal(st.evaluate(np.arange(4096, dtype=np.uint32)), want)


def test_near_compact_piece_square_rows(sp, oracle, net_blob, states):
    """Rows with at most 32 weights outside i8 take the 1 KiB path of the full-refresh kernel too: the u8 copy holds those
    weights clamped, the exact remainders are added from a side table (per-column sums through LDS). A net with 1-31, exactly
    32 and 33-60 such weights per row (conftest._near_rows_net; values up to the i16 extremes, several in one lane's columns)
    must equal the oracle bit for bit through the full refresh, the arena refresh and the incremental kernels (which read
    such rows from the i16 table)."""
    from conftest import NEAR_ROW_KIND

    st = states("near")
    assert st.compact_psq_rows == int((NEAR_ROW_KIND == 0).sum())
    assert st.near_psq_rows == int(((NEAR_ROW_KIND == 1) | (NEAR_ROW_KIND == 3)).sum())
    assert states("tame").near_psq_rows == 0
    pos = sp.random_positions(8192, seed=78, min_ply=0, max_ply=160, dfrc_every=4)
    mail, stm = sp.positions_to_mailboxes(pos)
    oracle.use(net_blob("near"), "near")
    want = oracle.eval_mailboxes(mail, stm)
    assert len(set(want.tolist())) > 2000
    assert np.array_equal(st.evaluate_once(pos), want)
    assert np.array_equal(st.evaluate_once(pos[:3]), want[:3])  # the tiny-batch route
    st.reserve_slots(3 * 8192)
    slots = np.arange(8192, dtype=np.uint32)
    st.reset(pos, slots)
    assert np.array_equal(st.evaluate(slots), want)
    # one ply of incremental updates on top (rebuilds of king-bucket changes go through the same full-refresh kernel)
    nxt, moved = sp.random_successors(pos, seed=5)
    idx = np.nonzero(moved)[0]
    got = st.update_evaluate(slots[idx], slots[idx] + 8192, nxt[idx])
    m2, s2 = sp.positions_to_mailboxes(nxt[idx])
    assert np.array_equal(got, oracle.eval_mailboxes(m2, s2))


@pytest.mark.parametrize("preset", ["tame", "extreme", "mixed", "near"])
def test_team_kernel_matches_the_wave_kernel(sp, oracle, net_blob, preset):
    """Full refreshes of at most 512 perspectives run one WORKGROUP per perspective (spx_ft_team_kernel: four waves fetch a
    quarter of the rows each, partial accumulators summed through LDS), larger ones one wave per perspective. Both kernels,
    forced onto the same batches through option ft_team_max, must agree with the oracle and with
    each other: evaluations, u8 activations, and the accumulators they leave in the arena."""
    def make(team_max):
        return sp.NnueState(sp.Network(net_blob(preset)), device=0, max_batch=4096, options={"ft_team_max": team_max})

    pos = sp.random_positions(3000, seed=404, min_ply=0, max_ply=160, dfrc_every=3)
    mail, stm = sp.positions_to_mailboxes(pos)
    oracle.use(net_blob(preset), preset)
    want = oracle.eval_mailboxes(mail, stm)
    with make(0) as wave, make(1 << 20) as team:
        for n in (1, 2, 63, 200, 256, 257, 3000):
            a, b = wave.evaluate_once(pos[:n]), team.evaluate_once(pos[:n])
            assert np.array_equal(a, want[:n]) and np.array_equal(b, want[:n]), n
            assert np.array_equal(wave.debug_ft(n), team.debug_ft(n)), n
        slots = np.arange(3000, dtype=np.uint32)
        for st in (wave, team):
            st.reserve_slots(3000)
            st.reset(pos, slots)
            assert np.array_equal(st.evaluate(slots), want)
    # the default context takes the team kernel up to 256 positions and the wave kernel beyond: covered by every other test


@pytest.mark.parametrize("preset", ["tame", "wild"])
def test_adjust_on_device_matches_reference_and_oracle(sp, oracle, net_blob, states, preset):
    """spx_adjust (device post-processing, rows a18 / f-4): golden staticEvalOnce / adjustEval<false> values of the
    compiled reference, then random positions with random correction-history terms against the oracle."""
    import json
    import os

    st = states(preset)
    golden = os.path.join(os.path.dirname(__file__), "golden", "adjust.jsonl")
    recs = [r for r in map(json.loads, open(golden)) if r["preset"] == preset]
    by_setting = {}
    for r in recs:
        by_setting.setdefault((tuple(r["contempt"]), tuple(r["optimism"])), []).append(r)
    for (contempt, optimism), group in by_setting.items():
        pos = sp.positions_from_fens([r["fen"] for r in group])
        raw = st.evaluate_once(pos)
        stat = st.adjust(pos, raw, contempt, optimism, stages=sp.ADJUST_STATIC)
        assert np.array_equal(stat, np.array([r["static"] for r in group], dtype=np.int32))
        want = np.array([r["adjusted"] for r in group], dtype=np.int32)
        assert np.array_equal(st.adjust(pos, stat, contempt, optimism, stages=sp.ADJUST_EVAL), want)
        assert np.array_equal(st.adjust(pos, raw, contempt, optimism), want)

    pos = sp.random_positions(3000, seed=31, min_ply=0, max_ply=200, dfrc_every=5)
    mail, stm = sp.positions_to_mailboxes(pos)
    raw = st.evaluate_once(pos)
    corr = np.random.default_rng(4).integers(-(1 << 20), 1 << 20, size=len(pos)).astype(np.int32)
    got = st.adjust(pos, raw, (13, -40), (200, -150), corrections=corr)
    want = [oracle.adjust(mail[i], stm[i], int(pos[i]["halfmove"]), int(raw[i]), (13, -40), (200, -150), 3, int(corr[i]))
            for i in range(len(pos))]
    assert np.array_equal(got, np.array(want, dtype=np.int32))
    assert st.adjust(pos[:0], raw[:0]).shape == (0,)


def test_contexts_are_reentrant_across_threads(sp, oracle, net_blob):
    """Lazy-SMP analogue (one NnueState per search thread, shared read-only Network - src/thread.h:147): contexts created
    from one net are used concurrently from different host threads (ctypes releases the GIL) without interfering."""
    import threading

    net = sp.Network(net_blob("wild"))
    batches = [sp.random_positions(3000 + 500 * t, seed=900 + t, min_ply=0, max_ply=150, dfrc_every=3) for t in range(4)]
    oracle.use(net_blob("wild"), "wild")
    wants = [oracle.eval_mailboxes(*sp.positions_to_mailboxes(b)) for b in batches]
    errors = []

    def worker(t):
        try:
            st = sp.NnueState(net, device=0, max_batch=8192)
            st.reserve_slots(len(batches[t]))
            for rep in range(25):
                got = st.evaluate_once(batches[t])
                if not np.array_equal(got, wants[t]):
                    errors.append((t, rep, "full"))
                    break
                slots = np.arange(len(batches[t]), dtype=np.uint32)
                st.reset(batches[t], slots)
                if not np.array_equal(st.evaluate(slots), wants[t]):
                    errors.append((t, rep, "arena"))
                    break
            st.close()
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_pipelined_async_calls_equal_the_synchronous_path(sp, oracle, net_blob):
    """spx_eval_full_device_async: several batches in flight on the context's two internal lanes (different inputs and
    sizes per call, incl. sort-free small ones) give exactly the results of the stream-ordered call. Inputs and outputs
    live in page-locked host memory (spx_host_alloc), which the device addresses directly."""
    import ctypes

    from stormphrax_amd import _lib

    lib = _lib.load()
    st = sp.NnueState(sp.Network(net_blob("wild")), device=0, max_batch=20000)
    sizes = [20000, 777, 12345, 1, 20000, 9000, 4096, 15000]
    batches = [sp.random_positions(n, seed=500 + i, min_ply=0, max_ply=150, dfrc_every=3) for i, n in enumerate(sizes)]
    wants = [st.evaluate_once(b) for b in batches]
    oracle.use(net_blob("wild"), "wild")
    assert np.array_equal(wants[1], oracle.eval_mailboxes(*sp.positions_to_mailboxes(batches[1])))
    ptrs, d_in, d_out = [], [], []
    try:
        for b in batches:
            pin, pout = lib.spx_host_alloc(len(b) * 32), lib.spx_host_alloc(len(b) * 4)
            assert pin and pout
            ptrs += [pin, pout]
            src = np.ctypeslib.as_array((ctypes.c_uint8 * (len(b) * 32)).from_address(pin))
            src[:] = b.view(np.uint8).reshape(-1)
            d_in.append(pin)
            d_out.append(np.ctypeslib.as_array((ctypes.c_int32 * len(b)).from_address(pout)))
        for rep in range(3):
            for out in d_out:
                out[:] = -1
            events = [st.evaluate_once_device_async(d_in[i], len(batches[i]), d_out[i].ctypes.data)
                      for i in range(len(batches))]
            assert all(events)
            st.synchronize()
            for i in range(len(batches)):
                assert np.array_equal(d_out[i], wants[i]), (rep, i)
        # the synchronous entry points still work on the same context afterwards
        assert np.array_equal(st.evaluate_once(batches[2]), wants[2])
        # a host buffer larger than the context's capacity is chunked over the same two lanes (copies of one chunk
        # overlap the evaluation of its neighbours)
        big = np.concatenate(batches * 2)
        assert len(big) > 4 * 20000
        assert np.array_equal(st.evaluate_once(big), np.concatenate(wants * 2))
    finally:
        st.close()
        for q in ptrs:
            lib.spx_host_free(q)


def _state_with_options(sp, blob, options, **kw):
    """A context with tuning knobs set (spx_ctx_set_option; none changes a result)."""
    return sp.NnueState(sp.Network(blob), device=0, options=options, **kw)


@pytest.mark.parametrize("preset", ["tame", "extreme", "mixed", "near", "realistic"])
def test_column_sliced_pipeline_equals_the_kernel_and_the_oracle(sp, oracle, net_blob, preset):
    """Big full refreshes take the column-sliced pipeline of spx_ftx.hip (extraction pass, counting sort by (king bucket, list
    length), plan, gather on the matrix pipe with the bucket's piece-square slab in LDS; the default from 16 384 positions up,
    here from 8 192: option ftx_min). Same sums mod 2^16, so the evaluations must equal those of spx_ft_kernel (a context created
    with sliced_ft=False: SPX_CTX_ONE_KERNEL_FT) and the oracle's bit for bit: batches at the pipeline's threshold,
    ragged ones, more than one pass (> 65 536 positions), nets with wide rows (low / high byte planes) and near-compact rows
    (taken as wide rows here), synchronous and pipelined calls."""
    blob = net_blob(preset)
    pos = sp.random_positions(70001, seed=909, min_ply=0, max_ply=160, dfrc_every=3)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=1 << 17, sliced_ft=False) as plain, \
            _state_with_options(sp, blob, {"ftx_min": 8200}, max_batch=1 << 17) as sliced:
        assert not plain.takes_sliced_pipeline(70001) and sliced.takes_sliced_pipeline(8200) and not sliced.takes_sliced_pipeline(8199)
        assert not plain.takes_sliced_pipeline(70001, pipelined=True) and sliced.takes_sliced_pipeline(8200, pipelined=True)
        want = plain.evaluate_once(pos)
        oracle.use(blob, preset)
        mail, stm = sp.positions_to_mailboxes(pos[:3000])
        assert np.array_equal(want[:3000], oracle.eval_mailboxes(mail, stm))
        for n in (70001, 8200, 20001, 65536, 65537, 100):  # (100: below the threshold, the one-kernel path of the same context)
            got = sliced.evaluate_once(pos[:n])
            bad = np.nonzero(got != want[:n])[0]
            assert bad.size == 0, f"n={n}: {bad.size} mismatches, first at {bad[0]}: {sp.position_to_fen(pos[bad[0]])}"
        # activations, byte for byte
        sliced.evaluate_once(pos[:9000])
        a = sliced.debug_ft(9000)
        plain.evaluate_once(pos[:9000])
        assert np.array_equal(a, plain.debug_ft(9000))
        # pipelined calls: preparation on the ring's streams, several batches in flight (inputs / outputs in page-locked host
        # memory the device addresses directly, as in test_pipelined_async_calls_equal_the_synchronous_path)
        import ctypes

        from stormphrax_amd import _lib

        lib = _lib.load()
        sizes, offs = (30000, 9000, 40000, 12345, 30000), (0, 1000, 2000, 3000, 4000)
        pin = lib.spx_host_alloc(len(pos) * 32)
        pouts = [lib.spx_host_alloc(n * 4) for n in sizes]
        assert pin and all(pouts)
        try:
            np.ctypeslib.as_array((ctypes.c_uint8 * (len(pos) * 32)).from_address(pin))[:] = pos.view(np.uint8).reshape(-1)
            outs = [np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(q)) for n, q in zip(sizes, pouts)]
            for rep in range(2):
                for o in outs:
                    o[:] = -1
                for o, off in zip(outs, offs):
                    assert sliced.evaluate_once_device_async(pin + 32 * off, len(o), o.ctypes.data)
                sliced.synchronize()
                for o, off in zip(outs, offs):
                    assert np.array_equal(o, want[off:off + len(o)]), (rep, off)
            # one pipelined call above a pass of the pipeline (65 536 positions): cut into passes that alternate between the lanes
            big = lib.spx_host_alloc(len(pos) * 4)
            assert big
            pouts.append(big)
            o = np.ctypeslib.as_array((ctypes.c_int32 * len(pos)).from_address(big))
            o[:] = -1
            assert sliced.evaluate_once_device_async(pin, len(pos), o.ctypes.data)
            sliced.synchronize()
            assert np.array_equal(o, want)
        finally:
            for q in [pin] + pouts:
                lib.spx_host_free(q)


@pytest.mark.parametrize("preset", ["tame", "realistic", "extreme"])
def test_hot_row_sets_do_not_change_results(sp, oracle, net_blob, preset):
    """The gather keeps a HOT SET of threat / pawn-pair rows in LDS beside the piece-square slab (spx_ftx.h; chosen from a histogram
    over the first big batch, or by spx_ctx_calibrate). A row is added from wherever it lives, so the set must not matter: the
    measured set, an empty one, the largest the tables hold, an adversarial random one and a deliberately BAD one (the rarest rows of
    the calibration batch) all give the evaluations of the one-kernel path and of the oracle - on the batch the set was measured
    on and on another one. (nnue_state.cpp:89-145, 309-354: the sums the reference adds up, in any order.)"""
    blob = net_blob(preset)
    pos = sp.random_positions(40000, seed=4242, min_ply=0, max_ply=160, dfrc_every=3)
    other = sp.random_positions(20000, seed=77, min_ply=8, max_ply=120, dfrc_every=4)
    oracle.use(blob, preset)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=1 << 16, sliced_ft=False) as plain:
        want, want_other = plain.evaluate_once(pos), plain.evaluate_once(other)
    mail, stm = sp.positions_to_mailboxes(pos[:2000])
    assert np.array_equal(want[:2000], oracle.eval_mailboxes(mail, stm))
    rng = np.random.default_rng(12)
    with _state_with_options(sp, blob, {"ftx_min": 1024, "tiny_batch_max": 0}, max_batch=1 << 16) as st:
        assert st.hot_rows().size == 0                       # nothing chosen before the first big batch
        assert np.array_equal(st.evaluate_once(pos), want)   # the call that measures the set
        measured = st.hot_rows()
        assert measured.size == 256 and len(set(measured.tolist())) == 256 and measured.max() < 64368
        assert np.array_equal(st.evaluate_once(other), want_other)
        assert np.array_equal(st.evaluate_once(pos[:1500]), want[:1500])
        sets = {"empty": np.zeros(0, dtype=np.uint32),
                "random": rng.choice(64368, 384, replace=False).astype(np.uint32),
                "one row": measured[:1],
                "reversed": measured[::-1].copy(),
                "pawn pairs only": np.arange(300, dtype=np.uint32)}
        for name, rows in sets.items():
            st.set_hot_rows(rows)
            assert np.array_equal(st.hot_rows(), rows), name
            for batch, ref in ((pos, want), (other, want_other), (pos[:1111], want[:1111])):
                got = st.evaluate_once(batch)
                bad = np.nonzero(got != ref)[0]
                assert bad.size == 0, f"{name}: {bad.size} mismatches, first {sp.position_to_fen(batch[bad[0]])}"
        # other sizes of the measured set (option ftx_hot_rows; the next big batch re-measures)
        for rows in (0, 64, 384):
            st.set_option("ftx_hot_rows", rows)
            assert np.array_equal(st.evaluate_once(other), want_other), rows
            assert st.hot_rows().size == rows
            assert np.array_equal(st.evaluate_once(pos), want), rows
        # pipelined calls on a freshly calibrated context
        st.set_option("ftx_hot_rows", 256)
        assert np.array_equal(_evaluate_through(sp, st, pos, "sliced_pipelined"), want)
        assert np.array_equal(_evaluate_through(sp, st, other, "sliced_pipelined"), want_other)


def test_pipeline_whose_launch_fails_falls_back_to_the_one_kernel_path(sp, net_blob):
    """a launch of the pipeline that the runtime refuses (simulated: option ftx_fail_launch - the first pass of a call,
    and the second pass of a call of three passes, after the first one has already written its part) must not surface as an error:
    the whole batch is served by spx_ft_kernel, the pipeline stays off for the context, later calls keep working."""
    blob = net_blob("wild")
    pos = sp.random_positions(150000, seed=912, min_ply=0, max_ply=160, dfrc_every=3)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=1 << 18, sliced_ft=False) as plain:
        want = plain.evaluate_once(pos)
    for fail_at, n in ((0, 40000), (1, 150000)):
        with _state_with_options(sp, blob, {}, max_batch=1 << 18) as st:
            assert st.takes_sliced_pipeline(n)
            assert np.array_equal(st.evaluate_once(pos[:n]), want[:n])  # (through the pipeline: scratch and hot set exist)
            st.set_option("ftx_fail_launch", fail_at)
            assert np.array_equal(st.evaluate_once(pos[:n]), want[:n]), fail_at
            assert not st.takes_sliced_pipeline(n)
            assert np.array_equal(st.evaluate_once(pos[:30000]), want[:30000])


def test_pipeline_that_does_not_fit_falls_back_to_the_one_kernel_path(sp, net_blob):
    """The column-sliced pipeline allocates its table and scratch sets on first use; when one does not fit (a context sized to
    fill the HBM: simulated with option ftx_fail_after) that call and all later ones take spx_ft_kernel - another GPU path, same
    results - while a gather already issued on the other lane finishes on the scratch set it has."""
    import ctypes

    from stormphrax_amd import _lib

    lib = _lib.load()
    blob = net_blob("tame")
    n = 40000
    pos = sp.random_positions(n, seed=911, min_ply=0, max_ply=160, dfrc_every=3)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=65536, sliced_ft=False) as plain:
        want = plain.evaluate_once(pos)
    pin = lib.spx_host_alloc(n * 32)
    pouts = [lib.spx_host_alloc(n * 4) for _ in range(4)]
    assert pin and all(pouts)
    try:
        np.ctypeslib.as_array((ctypes.c_uint8 * (n * 32)).from_address(pin))[:] = pos.view(np.uint8).reshape(-1)
        outs = [np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(q)) for q in pouts]
        for fail_after in (0, 1):  # (0: the first scratch set already; 1: the second lane's)
            with _state_with_options(sp, blob, {"ftx_fail_after": fail_after}, max_batch=65536) as st:
                assert st.takes_sliced_pipeline(n)
                for o in outs:
                    o[:] = -1
                for o in outs:
                    assert st.evaluate_once_device_async(pin, n, o.ctypes.data)
                st.synchronize()
                for o in outs:
                    assert np.array_equal(o, want), fail_after
                assert not st.takes_sliced_pipeline(n)
                assert np.array_equal(st.evaluate_once(pos), want)
    finally:
        for q in [pin] + pouts:
            lib.spx_host_free(q)


