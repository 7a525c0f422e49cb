"""Pins the CPU oracle (oracle/spx_oracle.c) against vectors produced by the COMPILED REFERENCE
(tests/golden/make_golden.py; Stormphrax 8.0.2 NnueState::evaluateOnce and its own feature indexers)."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_jsonl(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return [json.loads(line) for line in f]


@pytest.mark.parametrize("preset", ["tame", "wild", "extreme", "realistic"])
def test_oracle_matches_reference_evals(oracle, net_blob, preset):
    recs = load_jsonl("evals.jsonl")
    assert len(recs) >= 2000
    oracle.use(net_blob(preset), preset)
    bad = [(r["fen"], r[preset], got) for r in recs if (got := oracle.eval_fen(r["fen"])) != r[preset]]
    assert not bad, bad[:3]
    assert {r["src"] for r in recs} == {"startpos", "bench", "edge", "spx_random", "ref_playout"}


def test_oracle_matches_reference_feature_rows(oracle, net_blob, sp):
    """Row ids AND enumeration order equal the reference's (psq.h:338-365, threats.cpp:170-221)."""
    oracle.use(net_blob("tame"), "tame")
    recs = load_jsonl("features.jsonl")
    assert len(recs) >= 300
    buckets = set()
    for r in recs:
        pos = sp.positions_from_fens([r["fen"]])
        mail, stm = sp.positions_to_mailboxes(pos)
        assert int(stm[0]) == r["stm"]
        assert (int(np.count_nonzero(mail[0] != 12)) - 2) // 4 == r["bucket"]
        buckets.add(r["bucket"])
        for c in (0, 1):
            psq, thr = oracle.features(mail[0], c)
            assert psq.tolist() == r["psq"][c], r["fen"]
            assert thr.tolist() == r["thr"][c], r["fen"]
    assert buckets == set(range(8))  # every output bucket is covered


def test_golden_covers_king_buckets_and_mirroring():
    """The fixture exercises all 16 piece-square king buckets on both mirror halves (arch.h:53-65)."""
    seen = set()
    for r in load_jsonl("features.jsonl"):
        for c in (0, 1):
            seen.update(row // 704 for row in r["psq"][c])
    assert seen == set(range(16))


@pytest.mark.parametrize("preset", ["tame", "wild"])
def test_oracle_adjust_matches_reference(oracle, net_blob, sp, preset):
    """staticEvalOnce (contempt + clamp, eval.cpp:24-27,109-112) and adjustEval<false> (eval.cpp:30-67) of the compiled
    reference vs the restatement, incl. halfmove clocks beyond 200 and clamped scores."""
    recs = [r for r in load_jsonl("adjust.jsonl") if r["preset"] == preset]
    assert len(recs) > 1000
    oracle.use(net_blob(preset), preset)
    cache = {}
    for r in recs:
        fen = r["fen"]
        if fen not in cache:
            pos = sp.positions_from_fens([fen])
            mail, stm = sp.positions_to_mailboxes(pos)
            cache[fen] = (mail[0], int(stm[0]), int(fen.split()[4]), oracle.eval_fen(fen))
        mail, stm, halfmove, raw = cache[fen]
        stat = oracle.adjust(mail, stm, halfmove, raw, r["contempt"], r["optimism"], stages=1)
        assert stat == r["static"], r
        assert oracle.adjust(mail, stm, halfmove, stat, r["contempt"], r["optimism"], stages=2) == r["adjusted"], r
        # both stages in one call == adjustedStaticEval<false> (eval.cpp:81-91)
        assert oracle.adjust(mail, stm, halfmove, raw, r["contempt"], r["optimism"], stages=3) == r["adjusted"], r
