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


@pytest.mark.parametrize("preset", ["tame", "wild", "extreme", "realistic"])
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


def test_evaluate_costs_one_launch_whatever_the_pending_depth():
    """NnueState::evaluate materialises ALL pending plies in one spx_acc_update_chain_eval launch (round 2
    paid one ~29 us synchronous call per pending ply): the per-call time grows by a few microseconds per pending ply, not by
    a call per ply. Timings printed by the walk itself (evaluate_us_by_pending_plies: k=us(calls))."""
    fen = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
    out = subprocess.run([EXE, "--preset", "tame", "--walk", "7", "6000"] + fen.split(), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("evaluate_us_by_pending_plies:")][0]
    print(line)
    cost = {}
    for tok in line.split()[1:]:
        k, rest = tok.split("=")
        cost[int(k.rstrip("+"))] = (float(rest.split("(")[0]), int(rest.split("(")[1].rstrip(")")))
    assert cost[1][1] > 500 and 2 in cost and 3 in cost
    deep = [us for k, (us, calls) in cost.items() if k >= 3 and calls >= 20]
    assert deep and max(deep) < cost[1][0] + 40.0     # not (pending x one ~29 us call per ply: 3 pending cost 87 us in round 2)
    assert cost[1][0] < 80.0


@pytest.mark.parametrize("preset,seed,fen", [
    ("tame", 11, "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"),
    ("wild", 12, "bqnb1rkr/pp3ppp/3ppn2/2p5/5P2/P2P4/NPP1P1PP/BQ1BNRKR w HFhf - 2 9"),
    ("extreme", 13, "4k2r/1P4P1/8/3p4/4P3/8/p6p/R3K3 w Qk - 0 1"),
])
def test_apply_immediately_keeps_the_datagen_invariant(preset, seed, fen):
    """NnueState::applyImmediately (nnue_state.cpp:572-591) the way datagen uses it (datagen.cpp:257-262): every move made
    with the observer, its UpdateContext consumed at once, evaluate() == evaluateOnce() after every move, the stack depth
    unchanged - over games with castling (Chess960 too), en passant, promotions and king-bucket changes."""
    out = subprocess.run([EXE, "--preset", preset, "--datagen", str(seed), "1500"] + fen.split(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "1500 moves" in out.stdout and " 0 mismatches" in out.stdout
    print(out.stdout.strip())


# ---- the reference ENGINE on the GPU evaluator (oracle/ref_gpu.cpp) ----
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "sp_ref_gpu_tame")
needs_ref_gpu = pytest.mark.skipif(not os.path.exists(REF_GPU), reason="oracle/_ref/sp_ref_gpu_tame is built by `make -C oracle refgpu` "
                                   "where /root/reference exists and travels to the GPU box in oracle/_ref/")


def _ref_gpu(*args, stdin=None, timeout=900):
    out = subprocess.run([REF_GPU, *args], input=stdin, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def _bench_nodes(text):
    line = [ln for ln in text.splitlines() if ln.endswith(" nps") and " nodes " in ln and not ln.startswith("info")][-1]
    return int(line.split()[0])


@needs_ref_gpu
def test_reference_search_counts_the_same_nodes_on_gpu_evaluations():
    """Stormphrax's own objects (search, move ordering, TT, everything) linked against libspx_nnue.so, eval::NnueState's
    entry points forwarded to include/spx_nnue.hpp by link-time interposition - no reference source touched. The
    reference's only functional test is the node count of `bench` (src/bench.cpp:95-150: fixed-depth searches of 52
    positions): a search is deterministic in its evaluations, so the count with every evaluation made on the GPU (lazy
    accumulator stack: push / pop / evaluate with pending plies, null moves) must equal the CPU build's; in `both` mode every
    GPU value is also compared with the reference's own NnueState::evaluate on the spot."""
    both = _ref_gpu("bench", "4", "both")
    assert _bench_nodes(both) == _bench_nodes(_ref_gpu("bench", "4", "cpu")) == 129771
    report = [ln for ln in both.splitlines() if ln.startswith("# bench:")][-1]
    print(report)
    assert "checked against the CPU value: 0 mismatches" in report and "libspx_nnue (GPU)" in report
    assert int(report.split()[2]) > 50000  # evaluations
    gpu6, cpu6 = _ref_gpu("bench", "6", "gpu"), _ref_gpu("bench", "6", "cpu")
    assert _bench_nodes(gpu6) == _bench_nodes(cpu6) == 469287
    print([ln for ln in gpu6.splitlines() if ln.startswith("# bench:")][-1])


@needs_ref_gpu
def test_reference_raweval_and_datagen_invariant_on_the_gpu_state():
    """`raweval` (src/uci.cpp:797-800: eval::staticEvalOnce) of the golden FENs through the reference binary with the GPU
    evaluator behind it = the CPU build's output = the goldens; and datagen's step (src/datagen/datagen.cpp:257-262):
    NnueState::applyImmediately fed with the REFERENCE'S OWN UpdateContext (converted field by field to spx_move_delta,
    applied by spx_acc_update_observed) keeps staticEvalOnce(pos) == staticEval(pos, nnueState) over 1 500 plies of
    standard and double-Chess960 games."""
    recs = [json.loads(line) for line in open(os.path.join(GOLDEN, "evals.jsonl"))][:600]
    fens = "".join(r["fen"] + "\n" for r in recs)
    gpu = [int(v) for v in _ref_gpu("raweval", "gpu", stdin=fens).splitlines() if not v.startswith("#")]
    cpu = [int(v) for v in _ref_gpu("raweval", "cpu", stdin=fens).splitlines() if not v.startswith("#")]
    assert gpu == cpu == [max(-24999, min(24999, r["tame"])) for r in recs]
    out = _ref_gpu("game", "1500", "11")
    print(out.splitlines()[0])
    assert "broken 0 times on the GPU state; GPU value != CPU value 0 times" in out and "1500 plies" in out
