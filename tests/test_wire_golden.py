"""Wire formats and WDL normalisation against bytes / values produced by the COMPILED REFERENCE (tests/golden/pack.txt,
viri_games.txt, wdl.txt - written by oracle/ref_probe.cpp `pack` / `viri` / `wdl` through the reference's own
datagen::marlinformat::PackedBoard::pack, datagen::Viriformat and wdl::normalizeScore): SURVEY 8 rows f-2 and f-4.
CPU side: FEN packing, the host viriformat expander, the host move generator's move words and child records, the oracle's
and the library's WDL restatements. The same goldens drive the device kernels in tests/test_gpu_wire.py."""
import ctypes
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_pack():
    return [line.rstrip("\n").split(" | ") for line in open(os.path.join(GOLDEN, "pack.txt")) if not line.startswith("#")]


def load_games():
    """-> list of dict(plies=[(fen, uci, score, filtered, pack bytes)], stream=bytes)"""
    games, cur = [], None
    for line in open(os.path.join(GOLDEN, "viri_games.txt")):
        line = line.rstrip("\n")
        if line.startswith("GAME"):
            cur = {"plies": [], "stream": b""}
            games.append(cur)
        elif line.startswith("M "):
            fen, uci, score, filtered, hx = line[2:].split(" | ")
            cur["plies"].append((fen, uci, int(score), int(filtered), bytes.fromhex(hx)))
        elif line.startswith("V "):
            cur["stream"] = bytes.fromhex(line[2:])
        elif line.startswith("F "):  # datagen::Marlinformat's file for the same game: "<records kept> <hex>"
            parts = line.split(" ")
            cur["marlin_kept"], cur["marlin"] = int(parts[1]), bytes.fromhex(parts[2]) if len(parts) > 2 else b""
        elif line.startswith("T "):  # datagen::Fen's file, lines joined by ';'
            cur["fen_text"] = line[2:].replace(";", "\n")
    return games


def load_wdl():
    out = []
    for line in open(os.path.join(GOLDEN, "wdl.txt")):
        if line.startswith("#"):
            continue
        nums, fen = line.rstrip("\n").split(" | ")
        score, material, norm = (int(v) for v in nums.split())
        out.append((score, material, norm, fen))
    return out


def test_fen_packing_equals_the_reference_packed_board(sp):
    """spx_pos_from_fen == PackedBoard::pack(pos, 0) byte for byte: nibble order and colour bit, castling-rook code 6 (also
    for Chess960 rights), the relative en-passant square - present only while an en passant capture is LEGAL
    (Position::filterEp: pinned capturers, discovered rook checks along the rank, other checkers) - clocks, zero tail."""
    cases = load_pack()
    assert len(cases) > 900
    got = sp.positions_from_fens([fen for fen, _ in cases])
    bad = [fen for (fen, hx), rec in zip(cases, got) if rec.tobytes() != bytes.fromhex(hx)]
    assert not bad, bad[:5]
    assert sum(1 for _, hx in cases if bytes.fromhex(hx)[24] & 0x7F != 64) >= 8     # real ep squares in the set (the viriformat games add more)
    assert sum(1 for _, hx in cases if any(((b & 7) == 6) or ((b >> 4) & 7) == 6 for b in bytes.fromhex(hx)[8:24])) > 200


def test_viriformat_streams_of_the_reference_expand_to_its_own_records(sp):
    """The reference wrote these game streams (Viriformat::writeAllWithOutcome). The host expander must return, per played
    move, exactly the record the reference packs for the position before the move (score in `eval`), with the game's
    outcome byte in `wdl`, and the reference's filter decision (in check / noisy move, datagen.cpp:254)."""
    games = load_games()
    assert len(games) == 40 and sum(len(g["plies"]) for g in games) > 3000
    kinds = set()
    for g in games:
        stream, plies = g["stream"], g["plies"]
        assert len(stream) == 32 + 4 * len(plies) + 4 and stream[-4:] == b"\0\0\0\0"
        outcome = stream[30]
        # the start record: our packing of the start FEN + the outcome byte
        start = bytearray(sp.positions_from_fens([plies[0][0]])[0].tobytes())
        start[30] = outcome
        assert bytes(start) == stream[:32]
        records, n_games, keep = sp.viri_expand(stream, with_filter=True)
        assert n_games == 1 and len(records) == len(plies)
        for k, (fen, uci, score, filtered, packed) in enumerate(plies):
            want = bytearray(packed)
            want[30] = outcome
            assert records[k].tobytes() == bytes(want), (fen, uci)
            assert bool(keep[k]) == (not filtered), (fen, uci)
            word = int.from_bytes(stream[32 + 4 * k:34 + 4 * k], "little")
            assert int.from_bytes(stream[34 + 4 * k:36 + 4 * k], "little", signed=True) == score
            kinds.add(word >> 14)
            # the host move generator knows this move under the same 16-bit word, and makes the same next position
            moves, children, _ = sp.legal_moves(records[k])
            hit = np.nonzero(moves == word)[0]
            assert hit.size == 1, (fen, uci, hex(word))
            if k + 1 < len(plies):
                nxt = bytearray(plies[k + 1][4])
                nxt[28:30] = b"\0\0"  # the child record carries no score yet
                assert children[hit[0]].tobytes() == bytes(nxt), (fen, uci)
    assert kinds == {0, 1, 2, 3}  # normal, en passant, castling, promotion words all occur


def test_marlinformat_and_fen_outputs_equal_the_reference_writers(sp):
    """datagen's other two output formats (datagen.cpp:340-346). The probe pushed every golden game through
    datagen::Marlinformat and datagen::Fen as well; converting the VIRIFORMAT stream of a game must give the bytes / the text
    those writers produced (filter: datagen.cpp:254; records: marlinformat.cpp:32-57; lines: fen.cpp:32-66) - game by game,
    and for all games concatenated in one call."""
    from stormphrax_amd import _lib

    games = load_games()
    assert all("marlin" in g and "fen_text" in g for g in games)
    for g in games:
        records, n_games = sp.viri_to_marlinformat(g["stream"])
        assert n_games == 1 and len(records) == g["marlin_kept"] == len(g["marlin"]) // 32
        assert records.tobytes() == g["marlin"]
        text, n_games = sp.viri_to_fen(g["stream"])
        assert n_games == 1 and text == g["fen_text"]
    assert 0 < sum(g["marlin_kept"] for g in games) < sum(len(g["plies"]) for g in games)  # the filter dropped some, kept some
    everything = b"".join(g["stream"] for g in games)
    records, n_games = sp.viri_to_marlinformat(everything)
    assert n_games == 40 and records.tobytes() == b"".join(g["marlin"] for g in games)
    text, n_games = sp.viri_to_fen(everything)
    assert n_games == 40 and text == "".join(g["fen_text"] for g in games)
    assert all(line.rsplit(" | ", 1)[1] in ("0.0", "0.5", "1.0") for line in text.splitlines())
    # malformed input fails loudly: a truncated stream, an outcome byte that is none of the three
    with pytest.raises(_lib.SpxError):
        sp.viri_to_marlinformat(games[0]["stream"][:-3])
    broken = bytearray(games[0]["stream"])
    broken[30] = 7
    with pytest.raises(_lib.SpxError):
        sp.viri_to_fen(bytes(broken))
    assert sp.viri_to_marlinformat(b"")[0].shape == (0,) and sp.viri_to_fen(b"") == ("", 0)


def test_wdl_normalisation_restatements_match_the_reference(sp, oracle):
    """wdl::normalizeScore (f64 cubic + std::round) - the oracle's plain-C restatement and the library's SPX_HD function
    (the source the adjust kernel compiles) both reproduce the reference's integers; classicalMaterial likewise."""
    from stormphrax_amd import _lib

    lib = _lib.load()
    oracle.lib.spxo_wdl_normalize.argtypes = [ctypes.c_int32, ctypes.c_int32]
    oracle.lib.spxo_wdl_normalize.restype = ctypes.c_int32
    oracle.lib.spxo_classical_material.argtypes = [ctypes.c_void_p]
    oracle.lib.spxo_classical_material.restype = ctypes.c_int32
    rows = load_wdl()
    assert len(rows) > 3000
    fens = sorted({r[3] for r in rows})
    recs = dict(zip(fens, sp.positions_from_fens(fens)))
    mails = dict(zip(fens, sp.positions_to_mailboxes(np.array([recs[f] for f in fens], dtype=sp.PACKED_DTYPE))[0]))
    material, norm = ctypes.c_int32(), ctypes.c_int32()
    changed = 0
    for score, want_material, want, fen in rows:
        assert oracle.lib.spxo_classical_material(mails[fen].ctypes.data) == want_material
        assert oracle.lib.spxo_wdl_normalize(score, want_material) == want, (score, want_material)
        rec = np.ascontiguousarray(recs[fen]).reshape(1)
        assert lib.spx_debug_wdl(rec.ctypes.data, score, ctypes.byref(material), ctypes.byref(norm)) == 0
        assert (material.value, norm.value) == (want_material, want), (score, fen)
        changed += want != score
    assert changed > 2000  # decisive scores and zero pass through, the rest is rescaled
