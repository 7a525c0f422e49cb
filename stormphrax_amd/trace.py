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
    def __init__(self, path=None, lines=None):
        """`path`: a trace file (plain or .gz); or `lines`: the trace as an iterable of text lines."""
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

        import contextlib

        source = (contextlib.nullcontext(lines) if lines is not None
                  else gzip.open(path, "rt") if str(path).endswith(".gz") else open(path))
        with source as f:
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


class Forest:
    """Many recorded search trees as ONE tree under a virtual root, for spx_acc_replay_tree (BASELINE config 3 in the shape
    that thousands of concurrent searches give: VERDICT r3 item 4). Node 0 is the virtual root (the first tree's root
    position; never evaluated); tree t's root is a child of node 0 whose board has nothing to do with its parent's - the
    update kernels rebuild such a child from scratch, exactly what NnueState::reset does. Stored as a compressed .npz of flat
    arrays (tests/golden/forest_search_*.npz, written by tests/golden/make_golden.py `forest`): `parent` (node -> parent),
    `move` (6-byte UCI text per node: empty = a tree's root, "0000" = a null move), `root_fen`, `eval_node` / `eval_value`
    (the reference's NnueState::evaluate at those nodes)."""

    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.parent = z["parent"].astype(np.uint32)
        self.move = np.ascontiguousarray(z["move"])
        self.root_fens = [str(f) for f in z["root_fen"]]
        self.eval_node = z["eval_node"].astype(np.uint32)
        self.eval_value = z["eval_value"].astype(np.int32)
        self.depth = z["depth"].astype(np.int32)
        self.n_nodes = len(self.parent)
        self.n_trees = len(self.root_fens)

    @staticmethod
    def build(traces):
        """Flat arrays of a list of Trace objects (the recording side)."""
        parent, move, depth, eval_node, eval_value, fens = [0], [b""], [0], [], [], []
        for tr in traces:
            base = len(parent)
            fens.append(tr.root_fen)
            for k in range(tr.n_nodes):
                parent.append(0 if k == 0 else base + tr.parent[k])
                move.append(b"" if k == 0 else b"0000" if tr.null[k] else tr.moves[k].encode())
                depth.append(1 + tr.depth[k])
            eval_node += [base + e[0] for e in tr.evals]
            eval_value += [e[1] for e in tr.evals]
        return {"parent": np.asarray(parent, dtype=np.uint32), "move": np.asarray(move, dtype="S6"),
                "depth": np.asarray(depth, dtype=np.int16), "root_fen": np.asarray(fens),
                "eval_node": np.asarray(eval_node, dtype=np.uint32), "eval_value": np.asarray(eval_value, dtype=np.int32)}

    def positions(self):
        """Packed record of every node: one native call (spx_tree_expand_uci)."""
        from . import _lib

        roots = nnue.positions_from_fens(self.root_fens)
        all_roots = np.concatenate([roots[:1], roots])  # node 0, the virtual root, borrows the first tree's position
        out = np.zeros(self.n_nodes, dtype=nnue.PACKED_DTYPE)
        _lib.check(_lib.load().spx_tree_expand_uci(all_roots.ctypes.data, len(all_roots), self.parent.ctypes.data,
                                                    self.move.ctypes.data, self.n_nodes, out.ctypes.data))
        return out


def replay_forest(state, forest, positions=None):
    """Every tree of `forest` at once through ONE spx_acc_replay_tree call. -> (gpu values, reference values, device ms)"""
    pos = forest.positions() if positions is None else positions
    got, ms = state.replay_tree(pos, forest.parent, forest.eval_node)
    return got, forest.eval_value, ms
