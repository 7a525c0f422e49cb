"""The native host program on top of include/spx_nnue.hpp (the C++ mirror of eval::NnueState): `raweval`-style
evaluation of the reference's golden positions, and the accumulator STACK driven like a search (push / pop / evaluate
with lazily pending plies) under the reference's own invariant evaluate() == evaluateOnce()."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "stormphrax_amd", "spx_raweval")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def native_program():
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build(); hipcc is on the GPU box too
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "stormphrax_amd", "csrc")], stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


@pytest.mark.parametrize("preset", ["tame", "wild", "extreme"])
def test_raweval_matches_reference_goldens(preset):
    recs = [json.loads(line) for line in open(os.path.join(GOLDEN, "evals.jsonl"))]
    out = subprocess.run([EXE, "--preset", preset], input="".join(r["fen"] + "\n" for r in recs), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = [tuple(map(int, ln.split())) for ln in out.stdout.splitlines()]
    assert len(rows) == len(recs)
    assert [r[0] for r in rows] == [r[preset] for r in recs]                       # NnueState::evaluateOnce
    assert [r[1] for r in rows] == [max(-24999, min(24999, r[preset])) for r in recs]  # eval::staticEvalOnce, no contempt


@pytest.mark.parametrize("preset,seed,fen", [
    ("tame", 1, "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"),
    ("extreme", 2, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"),
    ("wild", 3, "bqnb1rkr/pp3ppp/3ppn2/2p5/5P2/P2P4/NPP1P1PP/BQ1BNRKR w HFhf - 2 9"),
    ("wild", 4, "4k2r/1P4P1/8/3p4/4P3/8/p6p/R3K3 w Qk - 0 1"),
])
def test_stack_walk_keeps_the_reference_invariant(preset, seed, fen):
    out = subprocess.run([EXE, "--preset", preset, "--walk", str(seed), "3000"] + fen.split(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
    words = out.stdout.replace(",", " ").split()  # "walk: N nodes, E evaluated, max depth D, 0 mismatches (...)"
    nodes, evaluated, depth = int(words[1]), int(words[3]), int(words[7])
    assert nodes == 3000 and evaluated > 1500 and depth >= 20
