"""Plain-Python restatement of the live fixed-node search of the self-play driver (SPX_SELFPLAY_SEARCH_NODES; the rules are
with SearchStepParams in stormphrax_amd/csrc/spx_kernels.h; the role is datagen's Searcher::runDatagenSearch,
src/search.cpp:212-239 with the soft node limit of src/datagen/datagen.cpp:78-80). TEST INFRASTRUCTURE: recursive, one node at
a time, nothing shared with the device's explicit-stack state machine (spx_search_step_kernel) except the rules:

  value(child)   = clamp(-network output of the child)                       (eval::adjustStatic, eval.cpp:24-27)
  search(node, depth, alpha, beta, ply)
                 = -(MATE - ply) / 0 without a legal move (check / stalemate); the best child value at depth 1; else the
                   children ordered by (value descending, viriformat move word ascending), fail-soft negamax, cut-off at
                   alpha >= beta
  root           : iterations 1, 2, ... over ONE expansion of the root (from iteration 2 on the previous best move first) until
                   the expansions reach the budget, the depth reaches LEVELS or the score is decisive; the move = the last
                   iteration's best, its score what the game loop sees

`verify_search_file` replays every recorded game: at every ply the restated search must pick the recorded move, and the
scores it returns, fed through tests/_datagen_rules.replay_game (datagen.cpp:213-300), must give the recorded scores, the
recorded length and the outcome byte. Leaf values come from the GPU's from-scratch evaluation of the children (what the
driver's incremental path must equal bit for bit), a sample of them is checked against the CPU oracle."""
import ctypes

import numpy as np

from _datagen_rules import (K_SCORE_WIN, VERIFICATION_SCORE_LIMIT, clamp_static, classical_material, parse_games,
                            replay_game)

INF, MATE, LEVELS = 32767, 32766, 8   # core.h:705-706; kSearchLevels


class Searcher:
    def __init__(self, sp, st, budget):
        self.sp, self.st, self.budget = sp, st, budget
        self.cache = {}
        self.nodes = 0          # expansions of the running search
        self.expanded = 0       # ... of all searches
        self.leaves = []        # (record bytes, raw eval) of everything evaluated, for the oracle sample

    def expand(self, rec):
        key = rec.tobytes()
        hit = self.cache.get(key)
        if hit is None:
            words, kids, in_check = self.sp.legal_moves(rec)
            values = []
            if len(words):
                raw = self.st.evaluate_once(kids)
                values = [clamp_static(-int(v)) for v in raw]
                if len(self.leaves) < 200000:
                    self.leaves.extend((kids[i].tobytes(), int(raw[i])) for i in range(0, len(words), 7))
            order = sorted(range(len(words)), key=lambda i: (-values[i], int(words[i])))
            hit = (words, kids, values, bool(in_check), order)
            self.cache[key] = hit
        self.nodes += 1
        self.expanded += 1
        return hit

    def search(self, rec, depth, alpha, beta, ply):
        words, kids, values, in_check, order = self.expand(rec)
        if len(words) == 0:
            return -(MATE - ply) if in_check else 0
        if depth == 1:
            return values[order[0]]
        best = -INF
        for i in order:
            v = -self.search(kids[i], depth - 1, -beta, -alpha, ply + 1)
            if v > best:
                best = v
            if v > alpha:
                alpha = v
            if alpha >= beta:
                break
        return best

    def root(self, rec):
        """-> (move word, score of the mover, child record) of the move the driver must play at `rec` (which has legal moves)."""
        self.cache.clear()
        self.nodes = 0
        words, kids, values, _, order = self.expand(rec)
        best_idx, best, depth = order[0], values[order[0]], 1
        while not (self.nodes >= self.budget or depth >= LEVELS or abs(best) > K_SCORE_WIN):
            depth += 1
            alpha, best, prev = -INF, -INF, best_idx
            for i in [prev] + [k for k in order if k != prev]:
                v = -self.search(kids[i], depth - 1, -INF, -alpha, 1)
                if v > best:
                    best, best_idx = v, i
                if v > alpha:
                    alpha = v
        return int(words[best_idx]), best, kids[best_idx], depth


def verify_search_file(sp, st, oracle, blob, max_plies, budget, tally=None):
    """-> (plies checked, nodes the restated searches expanded, deepest iteration seen). `oracle.use(...)` must have been
    called for the net `st` runs."""
    positions, n_games = sp.viri_expand(blob)
    games = parse_games(blob)
    assert len(games) == n_games and sum(len(g[1]) for g in games) == len(positions)
    wdl = oracle.lib.spxo_wdl_normalize
    wdl.argtypes, wdl.restype = [ctypes.c_int32, ctypes.c_int32], ctypes.c_int32

    def normalize(score, material):
        return int(wdl(int(score), int(material)))

    searcher = Searcher(sp, st, budget)
    start = checked = deepest = 0
    for gi, (_, moves, scores, outcome) in enumerate(games):
        n = len(moves)
        assert n >= 1
        before = positions[start:start + n]
        assert np.array_equal(before["eval"], scores) and np.all(before["wdl"] == outcome)
        mover = []
        last_child = None
        for k in range(n):
            word, score, child, depth = searcher.root(before[k])
            assert word == int(moves[k]), (gi, k, word, int(moves[k]))
            if k + 1 < n:
                assert child.tobytes()[:28] == before[k + 1].tobytes()[:28], (gi, k)
            mover.append(score)
            last_child = child
            deepest = max(deepest, depth)
            if k == 0:  # the first search doubles as the opening's verification search (datagen.cpp:176-190)
                white = not (int(before[0]["stm_ep"]) & 0x80)
                norm = normalize(score if white else -score, int(classical_material(before[:1])[0]))
                assert abs(norm) <= VERIFICATION_SCORE_LIMIT, (gi, norm)
        replies, _, in_check = sp.legal_moves(last_child)
        want_outcome, stop, recorded = replay_game(before, last_child, mover, normalize, max_plies,
                                                   len(replies) == 0 and bool(in_check), len(replies) > 0, tally)
        assert (want_outcome, stop) == (outcome, n), (gi, want_outcome, outcome, stop, n)
        assert recorded == [int(s) for s in scores], gi
        checked += n
        start += n
    # the leaves the restated searches saw (GPU, from scratch) against the CPU oracle
    sample = searcher.leaves[:: max(1, len(searcher.leaves) // 4096)]
    recs = np.frombuffer(b"".join(r for r, _ in sample), dtype=sp.PACKED_DTYPE)
    mail, stm = sp.positions_to_mailboxes(recs)
    assert np.array_equal(oracle.eval_mailboxes(mail, stm), np.array([v for _, v in sample])), "GPU evals differ from the CPU oracle"
    return checked, searcher.expanded, deepest
