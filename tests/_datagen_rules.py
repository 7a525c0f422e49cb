"""Plain-Python restatement of the game loop of the reference's data generator (src/datagen/datagen.cpp:176-300) for
checking recorded self-play games: opening verification, the win / loss / draw adjudication counters, Position::isDrawn
(src/position.cpp:603-667) and the terminal positions. TEST INFRASTRUCTURE: written from the reference's text, independent
of stormphrax_amd/csrc/spx_device_math.h (which the host path and the device step kernel share).

A game is replayed from what the file holds - the positions before every move (spx_viri_expand), the move words, the
recorded scores, the outcome byte - plus mover-point-of-view search scores supplied by the caller (for the depth-1 policy:
-staticEval(position after the move), from a from-scratch evaluation). `replay_game` returns the outcome the reference's loop
would have reached and the ply at which it would have stopped; the caller asserts both against the file."""
import numpy as np

K_SCORE_WIN = 25000           # core.h:708
VERIFICATION_SCORE_LIMIT = 500   # datagen.cpp:85
WIN_ADJ_MIN_SCORE = 1250      # datagen.cpp:87
DRAW_ADJ_MAX_SCORE = 10       # :88
DRAW_ADJ_MIN_PLIES = 70       # :90
WIN_ADJ_PLY_COUNT = 5         # :92
DRAW_ADJ_PLY_COUNT = 10       # :93
LIGHT_SQUARES = 0x55AA55AA55AA55AA  # bitboard.h:106
LOSS, DRAW, WIN = 0, 1, 2     # datagen/common.h:24-28 (white's point of view)

MATERIAL = np.array([1, 3, 3, 5, 9, 0, 5, 0], dtype=np.int64)  # Position::classicalMaterial (position.h:515-521); nibble 6 = rook with castling right


def clamp_static(v):
    """eval::adjustStatic with contempt 0 (eval.cpp:24-27)."""
    return max(-(K_SCORE_WIN - 1), min(K_SCORE_WIN - 1, int(v)))


def nibbles(rec):
    """Piece nibbles of a record in square order -> list of (square, type 0..5, is_white)."""
    occ = int(rec["occupancy"])
    out, idx = [], 0
    for sq in range(64):
        if occ >> sq & 1:
            nib = (int(rec["pieces"][idx >> 1]) >> (4 * (idx & 1))) & 0xF
            idx += 1
            t = nib & 7
            out.append((sq, 3 if t == 6 else t, not (nib & 8)))
    return out


def classical_material(positions):
    """Vectorised Position::classicalMaterial of packed records."""
    pieces = positions["pieces"].astype(np.int64)
    nib = np.stack([pieces & 0xF, pieces >> 4], axis=2).reshape(len(positions), 32)
    count = np.array([bin(int(o)).count("1") for o in positions["occupancy"]], dtype=np.int64)
    mask = np.arange(32)[None, :] < count[:, None]
    return (MATERIAL[nib & 7] * mask).sum(axis=1)


def ply_from_startpos(rec):
    """Position::plyFromStartpos (position.h:511-513)."""
    white = not (int(rec["stm_ep"]) & 0x80)
    return int(rec["fullmove"]) * 2 - (1 if white else 0) - 1


def identity(rec):
    """What the reference's Zobrist key distinguishes: placement incl. castling rights, side to move, en-passant square."""
    return (int(rec["occupancy"]), rec["pieces"].tobytes(), int(rec["stm_ep"]))


def insufficient_material(rec):
    """The material part of Position::isDrawn (position.cpp:639-666)."""
    minors = {True: [], False: []}
    for sq, t, white in nibbles(rec):
        if t in (0, 3, 4):
            return False
        if t in (1, 2):
            minors[white].append((sq, t))
    nb, nw = len(minors[False]), len(minors[True])
    if nb + nw == 0:
        return True                                   # KK
    if (nb == 0 and nw <= 1) or (nw == 0 and nb <= 1):
        return True                                   # KNK / KBK
    if nb <= 1 and nw <= 1 and all(t == 2 for _, t in minors[False] + minors[True]):   # KBKB "OCB", as the reference writes it
        def none_on_light(side):
            return not any(LIGHT_SQUARES >> sq & 1 for sq, _ in minors[side])
        return none_on_light(False) != none_on_light(True)
    return False


def is_drawn_by_repetition(key, history, halfmove):
    """Position::isDrawnByRepetition with ply = 0 (position.cpp:603-619): two earlier occurrences inside the window."""
    limit = max(0, len(history) - halfmove - 2)
    reps, i = 0, len(history) - 4
    while i >= limit:
        if history[i] == key:
            reps += 1
            if reps == 2:   # 1 + (ply < 0) with ply = 0 - 4
                return True
        i -= 2
    return False


def replay_game(before, after_last, mover_scores, normalize, max_plies, final_is_mate, final_has_moves, tally=None):
    """before[k]: record before move k (k < n); after_last: record after the last move; mover_scores[k]: the search score of
    move k from the mover's point of view (already clamped like a static eval); normalize(white_score, material) =
    wdl::normalizeScore. final_is_mate / final_has_moves describe after_last (checkmate? any legal move?).
    -> (outcome, plies played when the loop stopped, recorded scores the reference would have written).
    `tally` (optional dict) counts how games end: which branch of the rules stopped the loop."""
    n = len(before)

    def count(what):
        if tally is not None:
            tally[what] = tally.get(what, 0) + 1

    material = classical_material(before)
    win = loss = draw = 0
    history, recorded = [], []
    start_ply = ply_from_startpos(before[0])
    for k in range(n):
        pos = before[k]
        white = not (int(pos["stm_ep"]) & 0x80)
        white_score = int(mover_scores[k]) if white else -int(mover_scores[k])
        norm = normalize(white_score, int(material[k]))
        outcome = None
        # datagen.cpp:224-252 (isDecisive never holds for clamped static evals: only a search that found a mate gets there)
        if abs(white_score) > K_SCORE_WIN:
            outcome = WIN if white_score > 0 else LOSS
        elif norm > WIN_ADJ_MIN_SCORE:
            win, loss, draw = win + 1, 0, 0
        elif norm < -WIN_ADJ_MIN_SCORE:
            win, loss, draw = 0, loss + 1, 0
        elif start_ply + k >= DRAW_ADJ_MIN_PLIES and abs(norm) < DRAW_ADJ_MAX_SCORE:
            win, loss, draw = 0, 0, draw + 1
        else:
            win = loss = draw = 0
        if win >= WIN_ADJ_PLY_COUNT:
            outcome = WIN
        elif loss >= WIN_ADJ_PLY_COUNT:
            outcome = LOSS
        elif draw >= DRAW_ADJ_PLY_COUNT:
            outcome = DRAW
        history.append(identity(pos))
        nxt = before[k + 1] if k + 1 < n else after_last
        # Position::isDrawn(0, keyHistory) of the new position (datagen.cpp:264-268): overrides, score 0
        halfmove = int(nxt["halfmove"])
        why = None
        if halfmove >= 100:   # the 50-move branch returns before anything else is looked at: a draw unless checkmate
            assert k + 1 == n, "a game went on after its halfmove clock reached 100"
            why = None if final_is_mate else "fifty-move rule"
            if final_is_mate:
                count("checkmate on the 100th half-move (not a draw)")
        elif is_drawn_by_repetition(identity(nxt), history, halfmove):
            why = "threefold repetition"
        elif insufficient_material(nxt):
            why = "insufficient material"
        if why is None and k + 1 >= max_plies:   # the driver's own ply cap
            why = "ply cap"
        if why is not None:
            count("draw: " + why + (" (overriding an adjudication)" if outcome is not None else ""))
            recorded.append(0)
            return DRAW, k + 1, recorded
        recorded.append(0 if abs(white_score) <= 2 else white_score)
        if outcome is not None:
            count(("decisive score: " if abs(white_score) > K_SCORE_WIN else "adjudicated ") + ("white loss", "draw", "white win")[outcome])
            return outcome, k + 1, recorded
    # the loop went on: the position after the last recorded move must be terminal (datagen.cpp:213-221)
    assert not final_has_moves, "the recorded game stops although the reference's loop would have played on"
    last_white_to_move = not (int(after_last["stm_ep"]) & 0x80)
    if final_is_mate:
        count("checkmate")
        return (LOSS if last_white_to_move else WIN), n, recorded
    count("stalemate")
    return DRAW, n, recorded


def parse_games(blob):
    """viriformat stream -> list of (header bytes, move words, recorded scores, outcome byte)."""
    games, off = [], 0
    while off < len(blob):
        header = blob[off:off + 32]
        off += 32
        words = []
        while blob[off:off + 4] != b"\x00\x00\x00\x00":
            words.append(blob[off:off + 4])
            off += 4
        off += 4
        raw = np.frombuffer(b"".join(words), dtype=np.dtype([("move", "<u2"), ("score", "<i2")]))
        games.append((header, raw["move"].copy(), raw["score"].copy(), header[30]))
    return games


def verify_selfplay_file(sp, st, oracle, blob, max_plies, oracle_sample=4096, seed=0, tally=None):
    """Everything a recorded self-play file must satisfy:
    * every move legal (host expander) and the device expander identical to it;
    * the GPU's from-scratch evals of a sample of >= `oracle_sample` recorded positions equal the CPU oracle's;
    * every opening passes the reference's verification filter (datagen.cpp:176-190) under the depth-1 search;
    * every game replayed through the plain-Python restatement of datagen.cpp:213-300 ends where the file ends, with the
      file's outcome byte and the file's recorded scores (white point of view, |s| <= 2 -> 0, 0 on a drawn last move).
    `oracle.use(...)` must have been called for the net `st` runs. -> number of (game, ply) pairs checked."""
    import ctypes

    positions, n_games = sp.viri_expand(blob)
    games = parse_games(blob)
    assert len(games) == n_games and sum(len(g[1]) for g in games) == len(positions)
    device_records, device_games, bad = st.viri_expand(blob)
    assert (device_games, bad) == (n_games, 0) and device_records.tobytes() == positions.tobytes()
    full = st.evaluate_once(positions)
    # final positions (after the last recorded move) and the first positions' children, through the host move generator
    finals = np.zeros(n_games, dtype=sp.PACKED_DTYPE)
    final_has_moves, final_is_mate = [], []
    first_children, first_slices = [], []
    start = 0
    for gi, (_, moves, _, _) in enumerate(games):
        n = len(moves)
        assert n >= 1
        words, children, _ = sp.legal_moves(positions[start + n - 1])
        hit = np.nonzero(words == moves[-1])[0]
        assert len(hit) == 1
        finals[gi] = children[hit[0]]
        replies, _, in_check = sp.legal_moves(finals[gi])
        final_has_moves.append(len(replies) > 0)
        final_is_mate.append(len(replies) == 0 and in_check)
        _, kids, _ = sp.legal_moves(positions[start])
        first_slices.append((len(first_children), len(first_children) + len(kids)))
        first_children.extend(kids)
        start += n
    final_evals = st.evaluate_once(finals)
    first_evals = st.evaluate_once(np.array(first_children, dtype=sp.PACKED_DTYPE))
    # oracle: a sample of what the GPU evaluated from scratch
    rng = np.random.default_rng(seed)
    pool = np.concatenate([positions, finals])
    pool_evals = np.concatenate([full, final_evals])
    idx = rng.choice(len(pool), size=min(oracle_sample, len(pool)), replace=False)
    mail, stm = sp.positions_to_mailboxes(pool[idx])
    assert np.array_equal(oracle.eval_mailboxes(mail, stm), pool_evals[idx]), "GPU evals differ from the CPU oracle"
    wdl = oracle.lib.spxo_wdl_normalize
    wdl.argtypes, wdl.restype = [ctypes.c_int32, ctypes.c_int32], ctypes.c_int32

    def normalize(score, material):
        return int(wdl(int(score), int(material)))

    start = checked = 0
    for gi, (_, moves, scores, outcome) in enumerate(games):
        n = len(moves)
        before = positions[start:start + n]
        assert np.array_equal(before["eval"], scores) and np.all(before["wdl"] == outcome)
        nxt = np.concatenate([full[start + 1:start + n], final_evals[gi:gi + 1]])
        mover = np.array([clamp_static(-int(v)) for v in nxt])
        # opening verification: the depth-1 search of the first position, normalised (datagen.cpp:184-192)
        lo, hi = first_slices[gi]
        best = max(clamp_static(-int(v)) for v in first_evals[lo:hi])
        white = not (int(before[0]["stm_ep"]) & 0x80)
        norm_best = normalize(best if white else -best, int(classical_material(before[:1])[0]))
        assert abs(norm_best) <= VERIFICATION_SCORE_LIMIT, (gi, norm_best)
        want_outcome, stop, recorded = replay_game(before, finals[gi], mover, normalize, max_plies,
                                                   final_is_mate[gi], final_has_moves[gi], tally)
        assert (want_outcome, stop) == (outcome, n), (gi, want_outcome, outcome, stop, n)
        assert recorded == [int(s) for s in scores], gi
        checked += n
        start += n
    return checked
