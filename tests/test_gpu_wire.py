"""Device kernels against bytes / values produced by the COMPILED REFERENCE: the viriformat
expansion kernel, the move generator's child records and move words, and the WDL stage of the adjust kernel - same golden
files as tests/test_wire_golden.py."""
import ctypes

import numpy as np
import pytest

from test_wire_golden import load_games, load_wdl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def state(sp, net_blob):
    st = sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=8192)
    yield st
    st.close()


def test_device_viriformat_expansion_reproduces_the_reference_records(sp, state):
    games = load_games()
    blob = b"".join(g["stream"] for g in games)
    records, n_games, bad, keep = state.viri_expand(blob, with_filter=True)
    assert (n_games, bad) == (len(games), 0) and len(records) == sum(len(g["plies"]) for g in games)
    k = 0
    for g in games:
        outcome = g["stream"][30]
        for fen, uci, score, filtered, packed in g["plies"]:
            want = bytearray(packed)
            want[30] = outcome
            assert records[k].tobytes() == bytes(want), (fen, uci)
            assert bool(keep[k]) == (not filtered), (fen, uci)
            k += 1


def test_marlinformat_of_device_played_games(sp, state, tmp_path):
    """datagen's marlinformat / fen outputs of games the DEVICE played: the device expander's unfiltered records are the
    bytes of the host conversion (whose output is pinned to the reference writers in test_wire_golden.py), the reference's
    golden games give the same through the device expander, and the fen text has one line per record."""
    games = load_games()
    blob = b"".join(g["stream"] for g in games)
    records, _, _, keep = state.viri_expand(blob, with_filter=True)
    assert records[keep].tobytes() == b"".join(g["marlin"] for g in games)
    path = str(tmp_path / "games.vf")
    stats = state.selfplay(64, 200, out_path=path, max_plies=120, dfrc=True, seed=31)
    played = open(path, "rb").read()
    records, n_games, bad, keep = state.viri_expand(played, with_filter=True)
    assert (n_games, bad) == (stats["games"], 0) and len(records) == stats["positions"]
    marlin, g2 = sp.viri_to_marlinformat(played)
    assert g2 == n_games and records[keep].tobytes() == marlin.tobytes() and 0 < len(marlin) < len(records)
    text, _ = sp.viri_to_fen(played)
    lines = text.splitlines()
    assert len(lines) == len(marlin)
    for k in (0, len(lines) // 2, len(lines) - 1):
        fen, score, wdl = lines[k].split(" | ")
        assert fen == sp.position_to_fen(marlin[k]) and int(score) == int(marlin[k]["eval"])
        assert wdl == ("0.0", "0.5", "1.0")[int(marlin[k]["wdl"])]


def test_device_move_generator_writes_the_reference_child_records(sp, state):
    """For every position of the reference's games the kernel's children contain the move the reference played, under the
    reference's own 16-bit viriformat word, and the child's 32 bytes are PackedBoard::pack of the reference's next
    position: castling-rook codes dropped, en-passant square only when the capture is legal (Position::filterEp), clocks."""
    games = load_games()
    parents, words, nexts = [], [], []
    for g in games:
        stream, plies = g["stream"], g["plies"]
        for k in range(len(plies) - 1):
            rec = bytearray(plies[k][4])
            rec[28:30] = b"\0\0"
            parents.append(bytes(rec))
            words.append(int.from_bytes(stream[32 + 4 * k:34 + 4 * k], "little"))
            nxt = bytearray(plies[k + 1][4])
            nxt[28:30] = b"\0\0"
            nexts.append(bytes(nxt))
    pos = np.frombuffer(b"".join(parents), dtype=sp.PACKED_DTYPE)
    assert len(pos) > 3000
    for lo in range(0, len(pos), 2048):
        chunk = pos[lo:lo + 2048]
        out = state.movegen(chunk)
        for i in range(len(chunk)):
            first, count = int(out["first"][i]), int(out["count"][i])
            moves = out["moves"][first:first + count]
            hit = np.nonzero(moves == words[lo + i])[0]
            assert hit.size == 1, (sp.position_to_fen(chunk[i]), hex(words[lo + i]))
            assert out["children"][first + hit[0]].tobytes() == nexts[lo + i], sp.position_to_fen(chunk[i])


def test_wdl_stage_of_the_adjust_kernel_matches_the_reference(sp, state, oracle):
    """SPX_ADJUST_WDL == wdl::normalizeScore(score, classicalMaterial) of the compiled reference, integer for integer (f64
    on the device: the cubic, the division and std::round), and SPX_ADJUST_WHITE_POV | SPX_ADJUST_WDL == what
    runDatagenSearch hands to the adjudication counters (search.cpp:237-238)."""
    rows = load_wdl()
    scores = np.array([r[0] for r in rows], dtype=np.int32)
    want = np.array([r[2] for r in rows], dtype=np.int32)
    fens = [r[3] for r in rows]
    uniq = sorted(set(fens))
    recs = dict(zip(uniq, sp.positions_from_fens(uniq)))
    pos = np.array([recs[f] for f in fens], dtype=sp.PACKED_DTYPE)
    got = state.adjust(pos, scores, stages=sp.ADJUST_WDL)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {len(want)} differ"
    oracle.lib.spxo_wdl_normalize.argtypes = [ctypes.c_int32, ctypes.c_int32]
    oracle.lib.spxo_wdl_normalize.restype = ctypes.c_int32
    black = (pos["stm_ep"] & 0x80) != 0
    assert black.any() and (~black).any()
    white_pov = np.array([oracle.lib.spxo_wdl_normalize(int(-s if b else s), r[1]) for s, b, r in zip(scores, black, rows)],
                         dtype=np.int32)
    got = state.adjust(pos, scores, stages=sp.ADJUST_WHITE_POV | sp.ADJUST_WDL)
    assert np.array_equal(got, white_pov)
    assert np.array_equal(state.adjust(pos, scores, stages=sp.ADJUST_WHITE_POV), np.where(black, -scores, scores))
