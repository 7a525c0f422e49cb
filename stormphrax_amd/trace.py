"""Recorded make/unmake traces -> level-synchronous batches for the accumulator arena (BASELINE config 3).

Trace format (tests/golden/trace_*.txt, recorded from the compiled reference by tests/golden/make_golden.py):
    ROOT <fen>          the position NnueState::reset was called on
    PUSH <uci>          thread.applyMove: NnueState::push + Position::applyMove (src/thread.cpp:46-67)
    POP                 the guard's NnueState::pop (src/thread.h:116-122)
    EVAL <inc> <once>   NnueState::evaluate at the current node (lazy multi-ply update, nnue_state.cpp:636-697) and
                        evaluateOnce of the same position, as computed by the reference
Traces recorded from the reference's own alpha-beta SEARCH (`searchtrace`, tests/golden/trace_search_*.txt.gz) carry one
more field on PUSH and EVAL: the side to move (w / b) of the position the call acted on. A search also makes NULL moves,
which do not touch the NNUE stack (src/thread.cpp:28-44): below one, the side to move is the other one while the
accumulators are the parent's. The parser turns that into an explicit node - same board, side to move flipped, en-passant
square cleared - entered when a PUSH / EVAL names the other side and left when one names the stack's side again (or on the
POP that leaves its parent), so the replay sees an ordinary tree (a null node is an update with an empty delta).

The reference walks the tree depth-first with ONE accumulator stack. On the GPU every visited node gets an arena slot
and the tree is processed level by level: all (parent -> child) updates of depth d form one spx_acc_update batch.
"""
import numpy as np

from . import nnue


class Trace:
    def __init__(self, path):
        self.root_fen = None
        self.parent = [-1]          # node -> parent node (node 0 = root)
        self.depth = [0]
        self.moves = [None]
        self.evals = []             # (node, incremental value, evaluateOnce value) as recorded by the reference
        stack = [0]                 # node ids; a null-move node sits above the node the null move was made from
        self.null = [False]         # node -> entered by a null move (its record: the parent's with the side to move flipped)
        stm = [None]                # node -> side to move where known ('w' / 'b'), from the recorded fields
        import gzip

        def enter_null():
            self.parent.append(stack[-1])
            self.depth.append(len(stack))
            self.moves.append(None)
            self.null.append(True)
            stm.append({"w": "b", "b": "w"}.get(stm[stack[-1]]))
            stack.append(len(self.parent) - 1)

        def settle(side):
            """Make the top of the stack the position the recorded call acted on (side to move `side`)."""
            if side is None:
                return
            if stm[stack[-1]] is None:
                stm[stack[-1]] = side
            elif stm[stack[-1]] != side:
                if self.null[stack[-1]]:
                    stack.pop()       # back from the null move's subtree
                else:
                    enter_null()
                assert stm[stack[-1]] == side, "trace: side to move does not fit the stack"

        with (gzip.open(path, "rt") if str(path).endswith(".gz") else open(path)) as f:
            for line in f:
                t = line.split()
                if not t or t[0].startswith("#"):
                    continue
                if t[0] == "ROOT":
                    self.root_fen = " ".join(t[1:])
                    stm[0] = t[2] if len(t) > 2 else None
                elif t[0] == "PUSH":
                    settle(t[2] if len(t) > 2 else None)
                    self.parent.append(stack[-1])
                    self.depth.append(len(stack))
                    self.moves.append(t[1])
                    self.null.append(False)
                    stm.append({"w": "b", "b": "w"}.get(t[2]) if len(t) > 2 else None)
                    stack.append(len(self.parent) - 1)
                elif t[0] == "POP":
                    if self.null[stack[-1]]:
                        stack.pop()   # the null node does not occupy an NNUE stack entry
                    stack.pop()
                elif t[0] == "EVAL":
                    settle(t[3] if len(t) > 3 else None)
                    self.evals.append((stack[-1], int(t[1]), int(t[2])))
        self.n_nodes = len(self.parent)

    def positions(self):
        """Packed record of every node, by replaying the moves with the host chess core."""
        pos = np.zeros(self.n_nodes, dtype=nnue.PACKED_DTYPE)
        pos[0] = nnue.positions_from_fens([self.root_fen])[0]
        for node in range(1, self.n_nodes):  # nodes are numbered in visiting order: parents come first
            if self.null[node]:  # Position::applyNullMove: same board, other side to move, no en-passant square
                pos[node] = pos[self.parent[node]]
                pos[node]["stm_ep"] = ((pos[node]["stm_ep"] & 0x80) ^ 0x80) | 64
            else:
                pos[node] = nnue.apply_uci(pos[self.parent[node]], self.moves[node])
        return pos

    def levels(self):
        """[(parent_nodes, child_nodes)] per depth 1, 2, ... - each a batch of independent updates."""
        depth = np.asarray(self.depth)
        parent = np.asarray(self.parent)
        out = []
        for d in range(1, int(depth.max()) + 1):
            nodes = np.nonzero(depth == d)[0].astype(np.uint32)
            out.append((parent[nodes].astype(np.uint32), nodes))
        return out


def replay(state, trace, positions=None):
    """Materialise every node of `trace` in the arena (slot = node id) and evaluate the recorded EVAL nodes.
    Returns (gpu values, reference incremental values, reference evaluateOnce values)."""
    pos = trace.positions() if positions is None else positions
    state.reserve_slots(trace.n_nodes)
    state.reset(pos[:1], np.zeros(1, dtype=np.uint32))
    for parents, children in trace.levels():
        for lo in range(0, len(children), state.max_batch):
            hi = lo + state.max_batch
            state.update(parents[lo:hi], children[lo:hi], pos[children[lo:hi]])
    nodes = np.array([e[0] for e in trace.evals], dtype=np.uint32)
    got = np.concatenate([state.evaluate(nodes[lo:lo + state.max_batch]) for lo in range(0, len(nodes), state.max_batch)])
    return got, np.array([e[1] for e in trace.evals], dtype=np.int32), np.array([e[2] for e in trace.evals], dtype=np.int32)


def replay_native(state, trace, positions=None):
    """The same replay through spx_acc_replay_tree: one native call, levels on device-resident buffers, no host round trip
    per level. Returns (gpu values, reference incremental values, device milliseconds)."""
    pos = trace.positions() if positions is None else positions
    parents = np.asarray(trace.parent, dtype=np.int64)
    parents[0] = 0
    nodes = np.array([e[0] for e in trace.evals], dtype=np.uint32)
    got, ms = state.replay_tree(pos, parents.astype(np.uint32), nodes)
    return got, np.array([e[1] for e in trace.evals], dtype=np.int32), ms
