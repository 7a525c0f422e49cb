"""Host-side mirror of the reference's eval interface (src/eval/nnue.h:38-63, src/eval/nnue_state.h:85-116),
batched. Thin wrappers over the C ABI; numpy arrays in, numpy arrays out.

    net = Network.synthetic(preset="tame")            # eval::init() analogue (the default net is not available offline)
    state = NnueState(net, device=0, max_batch=65536)  # one per caller thread, like the reference's NnueState
    evals = state.evaluate_once(positions)             # == [NnueState::evaluateOnce(pos, pos.stm()) for pos in ...]
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import PackedPos, check

PACKED_DTYPE = np.dtype(
    [
        ("occupancy", "<u8"),
        ("pieces", "u1", (16,)),
        ("stm_ep", "u1"),
        ("halfmove", "u1"),
        ("fullmove", "<u2"),
        ("eval", "<i2"),
        ("wdl", "u1"),
        ("extra", "u1"),
    ]
)
assert PACKED_DTYPE.itemsize == 32

PRESETS = {"tame": 0, "wild": 1, "extreme": 2, "realistic": 3}
DEFAULT_SEED = 20260927


def synthetic_net_bytes(preset="tame", seed=DEFAULT_SEED):
    lib = _lib.load()
    n = lib.spx_synth_net_bytes()
    buf = np.empty(n, dtype=np.uint8)
    check(lib.spx_synth_net(seed, PRESETS[preset], buf.ctypes.data, n))
    return buf


ADJUST_STATIC, ADJUST_EVAL, ADJUST_WHITE_POV, ADJUST_WDL = 1, 2, 4, 8
CTX_WIDE_PSQ_ROWS = 1  # spx_ctx_create_ex flag: no compact (u8) copies of piece-square rows
CTX_SLICED_FT = 2       # spx_ctx_create_ex flag: big full refreshes through the column-sliced pipeline (spx_ftx.hip; the default)
CTX_ONE_KERNEL_FT = 4   # ... never: every full refresh through spx_ft_kernel


def adjust_params(contempt=(0, 0), optimism=(0, 0), stages=ADJUST_STATIC | ADJUST_EVAL):
    """spx_adjust_params with the reference's default tunables (tunable.h:161-169)."""
    params = _lib.AdjustParams()
    _lib.load().spx_adjust_defaults(ctypes.byref(params))
    params.contempt[0], params.contempt[1] = contempt
    params.optimism[0], params.optimism[1] = optimism
    params.stages = stages
    return params


class Network:
    """Immutable, shareable network (spx_net). Mirrors eval::init / getNetwork / defaultNetworkName."""

    def __init__(self, blob):
        lib = _lib.load()
        blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        handle = ctypes.c_void_p()
        check(lib.spx_net_load(blob.ctypes.data, blob.size, ctypes.byref(handle)))
        self._h = handle
        self.blob = blob

    @classmethod
    def synthetic(cls, preset="tame", seed=DEFAULT_SEED):
        return cls(synthetic_net_bytes(preset, seed))

    @property
    def name(self):
        return _lib.load().spx_net_name(self._h).decode()

    @property
    def digest(self):
        """FNV-1a 64 of the logical payload (identical for the plain and the zstd-compressed image of a net)."""
        return int(_lib.load().spx_net_digest(self._h))

    def psq_row_classes(self):
        """(rows that fit i8, rows with <= 32 weights outside i8, wide rows) of the 11 264 piece-square rows - how a context
        will serve them (1 KiB copy / 1 KiB copy + remainders / 2 KiB i16 row). Host-side."""
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        check(_lib.load().spx_net_psq_row_classes(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().spx_net_free(h)
            except Exception:  # interpreter shutdown: module globals may already be gone
                pass


class NnueState:
    """Device context (spx_ctx): weights resident on one GPU + scratch for `max_batch` positions."""

    CREATION_OPTIONS = ("scratch_cap", "compact_rows", "near_rows")  # shape what a context allocates: at creation only
    TEST_HOOKS = ("ftx_fail_after", "ftx_fail_launch")  # fault injection: dev library only (spx_debug_enable_test_hooks)

    def __init__(self, network, device=0, max_batch=65536, wide_psq_rows=False, sliced_ft=None, options=None):
        """options: {name: int} tuning knobs (include/spx_nnue.h); all of them travel in spx_ctx_create_opts' option string -
        nothing process-global is touched (the environment rewrite of round 5 was not thread-safe)."""
        lib = _lib.load()
        handle = ctypes.c_void_p()
        flags = (CTX_WIDE_PSQ_ROWS if wide_psq_rows else 0) | (0 if sliced_ft is None else CTX_SLICED_FT if sliced_ft else CTX_ONE_KERNEL_FT)
        options = dict(options or {})
        if any(k in options for k in self.TEST_HOOKS):
            check(lib.spx_debug_enable_test_hooks(1))
        text = ",".join(f"{k}={int(v)}" for k, v in options.items())
        check(lib.spx_ctx_create_opts(network._h, device, max_batch, flags, text.encode() if text else None, ctypes.byref(handle)))
        self._h = handle
        self._net = network
        # what one arena call (reset / update / evaluate) accepts: the scratch capacity (option scratch_cap, 4 Mi positions by
        # default) when the context was created for a larger full-refresh batch - trace.replay chunks by this
        self.max_batch = min(max_batch, int(lib.spx_ctx_scratch_batch(handle)))
        self.call_limit = max_batch

    def set_option(self, name, value):
        """spx_ctx_set_option: one tuning knob of this context (never changes a result)."""
        if name in self.TEST_HOOKS:
            check(_lib.load().spx_debug_enable_test_hooks(1))
        check(_lib.load().spx_ctx_set_option(self._h, name.encode(), int(value)))

    def ftx_walk(self, slot=-1):
        """spx_debug_ftx_walk: what the last packed walk of a scratch set holds -> dict. `global_steps` / `lds_steps` are per column
        slice (the high-byte planes' steps differ per slice - an XCD drops the planes that are all zero in its slice -: their sum
        over the 8 slices / 8), `global_rows` / `lds_rows` in rows of 1 KiB (8 slices of 128 B)."""
        out = np.zeros(8, dtype=np.uint32)
        check(_lib.load().spx_debug_ftx_walk(self._h, slot, out.ctypes.data))
        groups, stages, cold_steps, lds_steps, cold_rows, lds_rows, hi_steps_all, hi_slices = (int(v) for v in out)
        return {"groups": groups, "stages": stages, "global_steps": cold_steps + hi_steps_all / 8.0, "lds_steps": lds_steps,
                "global_rows": cold_rows + hi_slices / 8.0, "lds_rows": lds_rows, "cold_steps": cold_steps, "cold_rows": cold_rows,
                "high_plane_steps_all_slices": hi_steps_all, "high_plane_slices_fetched": hi_slices}

    def ftx_lists(self, n, slot=-1):
        """spx_debug_ftx_lists: the extraction pass's row lists of the last batch (n positions) in the net's row numbering ->
        per perspective 2 i + c (c = colour, 1 = white) a tuple (piece-square rows, threat / pawn-pair rows, piece-square rows with a high plane listed)."""
        counts = np.zeros((2 * n, 3), dtype=np.uint32)
        rows = np.zeros((2 * n, 576), dtype=np.uint32)
        check(_lib.load().spx_debug_ftx_lists(self._h, slot, n, counts.ctypes.data, rows.ctypes.data))
        out = []
        for q in range(2 * n):
            a, b, c = (int(v) for v in counts[q])
            out.append((rows[q, :a].copy(), rows[q, a:a + b].copy(), rows[q, a + b:a + b + c].copy()))
        return out

    def calibrate(self, d_positions_ptr, n):
        """spx_ctx_calibrate: choose the gather's hot set (the threat / pawn-pair rows kept in LDS) from a device-resident batch."""
        check(_lib.load().spx_ctx_calibrate(self._h, d_positions_ptr, n))

    def set_hot_rows(self, rows):
        """spx_ctx_set_hot_rows (test entry point): the hot set given instead of measured; rows = distinct ids < 64 368."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        check(_lib.load().spx_ctx_set_hot_rows(self._h, rows.ctypes.data if rows.size else None, rows.size))

    def hot_rows(self):
        """The hot set in slot order (empty before the first calibration)."""
        out = np.empty(1024, dtype=np.uint32)
        n = ctypes.c_size_t()
        check(_lib.load().spx_ctx_get_hot_rows(self._h, out.ctypes.data, out.size, ctypes.byref(n)))
        return out[: n.value].copy()

    def evaluate_once(self, positions):
        """Batched NnueState::evaluateOnce: packed positions (PACKED_DTYPE array) -> int32 raw evals (stm view)."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        out = np.empty(pos.shape[0], dtype=np.int32)
        check(_lib.load().spx_eval_full(self._h, pos.ctypes.data, pos.shape[0], out.ctypes.data))
        return out

    def evaluate_once_device(self, d_positions_ptr, n, d_out_ptr, stream_ptr=None):
        """Device-resident variant: raw device pointers (e.g. torch tensors' data_ptr()), enqueued on `stream_ptr`."""
        check(_lib.load().spx_eval_full_device(self._h, d_positions_ptr, n, d_out_ptr, stream_ptr))

    def adjust(self, positions, evals, contempt=(0, 0), optimism=(0, 0), stages=ADJUST_STATIC | ADJUST_EVAL,
               corrections=None, params=None):
        """eval::adjustStatic (+ contempt[stm], clamp) and/or eval::adjustEval (material scaling, optimism, halfmove
        damping, optional correction / 2048, clamp) of raw evals on the device -> new int32 array.
        contempt / optimism are indexed by colour (black, white) like eval::Contempt / eval::Optimism."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        out = np.array(evals, dtype=np.int32, copy=True)
        assert pos.shape[0] == out.shape[0]
        if params is None:
            params = adjust_params(contempt, optimism, stages)
        corr = None if corrections is None else np.ascontiguousarray(corrections, dtype=np.int32)
        check(_lib.load().spx_adjust(self._h, pos.ctypes.data, pos.shape[0], ctypes.byref(params),
                                     None if corr is None else corr.ctypes.data, out.ctypes.data))
        return out

    def viri_expand(self, blob, with_filter=False):
        """viriformat game stream -> one record per played move, replayed on the device (spx_viri_expand_gpu).
        -> (records, games, bad_games[, unfiltered mask: what the reference's marlinformat output would keep])."""
        data = np.frombuffer(blob, dtype=np.uint8)
        n, games, bad = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        lib = _lib.load()
        check(lib.spx_viri_expand_gpu(self._h, data.ctypes.data, data.size, None, None, 0, ctypes.byref(n),
                                      ctypes.byref(games), ctypes.byref(bad)))
        out = np.zeros(n.value, dtype=PACKED_DTYPE)
        keep = np.zeros(n.value, dtype=np.uint8) if with_filter else None
        check(lib.spx_viri_expand_gpu(self._h, data.ctypes.data, data.size, out.ctypes.data,
                                      None if keep is None else keep.ctypes.data, n.value, ctypes.byref(n),
                                      ctypes.byref(games), ctypes.byref(bad)))
        return (out, games.value, bad.value, keep.astype(bool)) if with_filter else (out, games.value, bad.value)

    def movegen(self, positions, parent_values=None, capacity=None):
        """Legal moves + child records of every position, generated on the device (spx_movegen).
        -> dict(children, moves, parents, first, count, in_check)."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        n = pos.shape[0]
        capacity = capacity if capacity is not None else 64 * n + 256
        children = np.zeros(capacity, dtype=PACKED_DTYPE)
        moves = np.zeros(capacity, dtype=np.uint16)
        parents = np.zeros(capacity, dtype=np.uint32)
        first = np.zeros(n, dtype=np.uint32)
        count = np.zeros(n, dtype=np.uint32)
        in_check = np.zeros(n, dtype=np.uint8)
        pv = None if parent_values is None else np.ascontiguousarray(parent_values, dtype=np.uint32)
        total = ctypes.c_size_t()
        check(_lib.load().spx_movegen(self._h, pos.ctypes.data, n, None if pv is None else pv.ctypes.data,
                                      children.ctypes.data, moves.ctypes.data, parents.ctypes.data, first.ctypes.data,
                                      count.ctypes.data, in_check.ctypes.data, capacity, ctypes.byref(total)))
        t = total.value
        return {"children": children[:t], "moves": moves[:t], "parents": parents[:t], "first": first, "count": count,
                "in_check": in_check.astype(bool)}

    @property
    def scratch_batch(self):
        """Positions per internal chunk (spx_ctx_scratch_batch): larger spx_eval_full* calls are walked in chunks."""
        return int(_lib.load().spx_ctx_scratch_batch(self._h))

    @property
    def compact_psq_rows(self):
        """Piece-square rows (of 11264) this context serves from their 1 KiB u8 copy (all weights fit i8)."""
        return int(_lib.load().spx_ctx_compact_psq_rows(self._h))

    @property
    def near_psq_rows(self):
        """Piece-square rows with at most 32 weights outside i8: 1 KiB copy + exact remainders in the full-refresh kernel."""
        return int(_lib.load().spx_ctx_near_psq_rows(self._h))

    def count_rows(self, positions):
        """(psq rows fetched wide (2 KiB), psq rows fetched compact (1 KiB), threat / pawn-pair rows (1 KiB)) a full
        refresh of the batch gathers through THIS context, both perspectives summed. Host-side count."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        check(_lib.load().spx_ctx_count_rows(self._h, pos.ctypes.data, pos.shape[0], ctypes.byref(a), ctypes.byref(b),
                                             ctypes.byref(c)))
        return a.value, b.value, c.value

    def random_positions_device(self, d_out_ptr, count, seed=1, min_ply=8, max_ply=120, dfrc_every=4):
        """spx_random_positions_gpu: `count` seeded random-playout records written to device memory at d_out_ptr."""
        check(_lib.load().spx_random_positions_gpu(self._h, seed, count, min_ply, max_ply, dfrc_every, d_out_ptr))

    def evaluate_once_device_async(self, d_positions_ptr, n, d_out_ptr):
        """Pipelined variant (spx_eval_full_device_async): returns the hipEvent_t handle that marks the batch done."""
        ev = ctypes.c_void_p()
        check(_lib.load().spx_eval_full_device_async(self._h, d_positions_ptr, n, d_out_ptr, ctypes.byref(ev)))
        return ev.value

    def synchronize(self):
        check(_lib.load().spx_ctx_synchronize(self._h))

    def takes_sliced_pipeline(self, n, pipelined=False):
        """Does a full refresh of n positions take the column-sliced pipeline (spx_ftx.hip) on this context - as a stream-ordered
        call, or (pipelined=True) as a call of evaluate_once_device_async, whose threshold is lower?"""
        return bool(_lib.load().spx_ctx_sliced_ft(self._h, n) & (2 if pipelined else 1))

    def profile_begin(self, max_calls):
        check(_lib.load().spx_profile_begin(self._h, max_calls))

    def profile_end(self):
        """-> (sort_ms_total, ft_kernel_ms_total, mlp_kernel_ms_total, calls) since profile_begin."""
        s, a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t()
        check(_lib.load().spx_profile_end(self._h, ctypes.byref(s), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return s.value, a.value, b.value, c.value

    def profile_prepare_ms(self):
        """Of the calls the last profile_end summed up: the time between the sorts and the FT stage's main kernel (the column-sliced
        pipeline's preparation kernels on stream-ordered calls)."""
        v = ctypes.c_double()
        check(_lib.load().spx_profile_last_prepare_ms(self._h, ctypes.byref(v)))
        return v.value

    # ---- incremental path: accumulator arena (mirrors NnueState::reset / push+applyMove / evaluate) ----
    def reserve_slots(self, n_slots):
        check(_lib.load().spx_acc_reserve(self._h, n_slots))

    def reset(self, positions, slots):
        """NnueState::reset for each (position, slot): full refresh into the arena."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        assert pos.shape[0] == slots.shape[0]
        check(_lib.load().spx_acc_refresh(self._h, pos.ctypes.data, slots.ctypes.data, pos.shape[0]))

    def update(self, parent_slots, child_slots, child_positions):
        """One ply of incremental updates for independent (parent slot -> child slot) pairs."""
        pos = np.ascontiguousarray(child_positions, dtype=PACKED_DTYPE)
        ps = np.ascontiguousarray(parent_slots, dtype=np.uint32)
        cs = np.ascontiguousarray(child_slots, dtype=np.uint32)
        assert pos.shape[0] == ps.shape[0] == cs.shape[0]
        check(_lib.load().spx_acc_update(self._h, ps.ctypes.data, cs.ctypes.data, pos.ctypes.data, pos.shape[0]))

    def update_evaluate(self, parent_slots, child_slots, child_positions):
        """update() followed by evaluate(child_slots) in one fused call (push + applyMove + evaluate).
        child_slots=None: eval-only children - evaluated from registers, nothing stored in the arena."""
        pos = np.ascontiguousarray(child_positions, dtype=PACKED_DTYPE)
        ps = np.ascontiguousarray(parent_slots, dtype=np.uint32)
        cs = None if child_slots is None else np.ascontiguousarray(child_slots, dtype=np.uint32)
        out = np.empty(pos.shape[0], dtype=np.int32)
        check(_lib.load().spx_acc_update_eval(self._h, ps.ctypes.data, None if cs is None else cs.ctypes.data,
                                              pos.ctypes.data, pos.shape[0], out.ctypes.data))
        return out

    def update_observed(self, parent_slots, child_slots, child_positions, deltas, evaluate=True):
        """Incremental update from host-captured observer deltas (a ctypes array of MoveDelta); optional evaluation."""
        pos = np.ascontiguousarray(child_positions, dtype=PACKED_DTYPE)
        ps = np.ascontiguousarray(parent_slots, dtype=np.uint32)
        cs = np.ascontiguousarray(child_slots, dtype=np.uint32)
        out = np.empty(pos.shape[0], dtype=np.int32) if evaluate else None
        check(_lib.load().spx_acc_update_observed(self._h, ps.ctypes.data, cs.ctypes.data, pos.ctypes.data,
                                                  ctypes.addressof(deltas), pos.shape[0],
                                                  out.ctypes.data if evaluate else None))
        return out

    def replay_tree(self, positions, parents, eval_nodes):
        """spx_acc_replay_tree: the whole recorded tree level by level on the device -> (values at eval_nodes, gpu ms)."""
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        par = np.ascontiguousarray(parents, dtype=np.uint32)
        nodes = np.ascontiguousarray(eval_nodes, dtype=np.uint32)
        out = np.empty(nodes.shape[0], dtype=np.int32)
        ms = ctypes.c_double()
        check(_lib.load().spx_acc_replay_tree(self._h, pos.ctypes.data, par.ctypes.data, pos.shape[0], nodes.ctypes.data,
                                              nodes.shape[0], out.ctypes.data, ctypes.byref(ms)))
        return out, ms.value

    def evaluate(self, slots):
        """NnueState::evaluate on materialised slots -> int32 raw evals (side to move of each slot's position)."""
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.empty(slots.shape[0], dtype=np.int32)
        check(_lib.load().spx_acc_eval(self._h, slots.ctypes.data, slots.shape[0], out.ctypes.data))
        return out

    def selfplay(self, n_games, target_games, out_path=None, max_plies=300, dfrc=False, temperature_cp=30, seed=1,
                 host_threads=0, host_movegen=False, search_nodes=0):
        """Batched self-play (config 4 shape); returns the stats dict. See spx_selfplay_run.
        host_movegen=True generates moves with the host chess core instead of the device kernel.
        search_nodes=k: a live fixed-node search of k expanded nodes picks every move (SPX_SELFPLAY_SEARCH_NODES; 0 = the
        depth-1 policy; k = 1 plays the same games through the search driver); stats["steps"] then counts expanded nodes."""
        params = _lib.SelfplayParams(n_games, target_games, max_plies, 0, int(dfrc), temperature_cp, host_threads,
                                     (1 if host_movegen else 0) | (int(search_nodes) << 8), seed)
        stats = _lib.SelfplayStats()
        check(_lib.load().spx_selfplay_run(self._h, ctypes.byref(params), out_path.encode() if out_path else None,
                                           ctypes.byref(stats)))
        return {"games": stats.games, "positions": stats.positions, "evals": stats.evals, "steps": stats.steps,
                "outcomes": list(stats.outcomes), "seconds": stats.seconds, "gpu_seconds": stats.gpu_seconds}

    def debug_ft(self, n):
        out = np.empty((n, 1024), dtype=np.uint8)
        check(_lib.load().spx_debug_copy_ft(self._h, n, out.ctypes.data))
        return out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().spx_ctx_destroy(h)
            except Exception:  # interpreter shutdown
                pass

    def __del__(self):
        self.close()


class DeviceGroup:
    """spx_group: one context per GPU inside this process; a batch is cut into contiguous shards, one per member, each
    evaluated on its own host thread (SURVEY 8e for a native host; the harnesses' multi-GPU runs use one process per GPU,
    stormphrax_amd/distributed.py). `devices` = HIP ordinals, None = every visible device; an ordinal may repeat."""

    def __init__(self, network, devices=None, max_batch_per_device=65536, wide_psq_rows=False):
        lib = _lib.load()
        handle = ctypes.c_void_p()
        ids = [] if devices is None else [int(d) for d in devices]
        arr = (ctypes.c_int * len(ids))(*ids) if ids else None
        flags = CTX_WIDE_PSQ_ROWS if wide_psq_rows else 0
        check(lib.spx_group_create(network._h, arr, len(ids), max_batch_per_device, flags, ctypes.byref(handle)))
        self._h = handle
        self._net = network

    def __len__(self):
        return int(_lib.load().spx_group_size(self._h))

    def shard(self, n, index):
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        check(_lib.load().spx_group_shard(self._h, n, index, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def evaluate_once(self, positions):
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        out = np.empty(pos.shape[0], dtype=np.int32)
        check(_lib.load().spx_group_eval_full(self._h, pos.ctypes.data, pos.shape[0], out.ctypes.data))
        return out

    def adjust(self, positions, evals, contempt=(0, 0), optimism=(0, 0), stages=ADJUST_STATIC | ADJUST_EVAL,
               corrections=None):
        pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
        out = np.array(evals, dtype=np.int32, copy=True)
        assert pos.shape[0] == out.shape[0]
        params = adjust_params(contempt, optimism, stages)
        corr = None if corrections is None else np.ascontiguousarray(corrections, dtype=np.int32)
        check(_lib.load().spx_group_adjust(self._h, pos.ctypes.data, pos.shape[0], ctypes.byref(params),
                                           None if corr is None else corr.ctypes.data, out.ctypes.data))
        return out

    def selfplay(self, n_games, target_games, out_path=None, max_plies=300, dfrc=False, temperature_cp=30, seed=1,
                 search_nodes=0):
        """spx_group_selfplay_run: the games dealt to the members, one host thread and one device each; output files
        <out_path>.<member>.vf; summed stats."""
        params = _lib.SelfplayParams(n_games, target_games, max_plies, 0, int(dfrc), temperature_cp, 0,
                                     int(search_nodes) << 8, seed)
        stats = _lib.SelfplayStats()
        check(_lib.load().spx_group_selfplay_run(self._h, ctypes.byref(params), out_path.encode() if out_path else None,
                                                 ctypes.byref(stats)))
        return {"games": stats.games, "positions": stats.positions, "evals": stats.evals, "steps": stats.steps,
                "outcomes": list(stats.outcomes), "seconds": stats.seconds, "gpu_seconds": stats.gpu_seconds}

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().spx_group_destroy(h)
            except Exception:  # interpreter shutdown
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def device_count():
    """Visible HIP devices (0 when there is none - every evaluation entry point then fails with SPX_ERR_NO_DEVICE)."""
    n = ctypes.c_int(0)
    _lib.load().spx_device_count(ctypes.byref(n))
    return n.value


# ---- position plumbing ----
def positions_from_fens(fens):
    lib = _lib.load()
    out = np.zeros(len(fens), dtype=PACKED_DTYPE)
    for i, fen in enumerate(fens):
        check(lib.spx_pos_from_fen(fen.encode(), out[i : i + 1].ctypes.data))
    return out


def position_to_fen(rec):
    buf = ctypes.create_string_buffer(128)
    rec = np.ascontiguousarray(rec, dtype=PACKED_DTYPE).reshape(1)
    check(_lib.load().spx_pos_to_fen(rec.ctypes.data, buf, 128))
    return buf.value.decode()


def positions_to_mailboxes(positions):
    lib = _lib.load()
    pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
    mail = np.empty((pos.shape[0], 64), dtype=np.uint8)
    stm = np.empty(pos.shape[0], dtype=np.uint8)
    s = ctypes.c_int()
    for i in range(pos.shape[0]):
        check(lib.spx_pos_to_mailbox(pos[i : i + 1].ctypes.data, mail[i].ctypes.data, ctypes.byref(s)))
        stm[i] = s.value
    return mail, stm


def apply_uci(rec, uci):
    """Position::applyMove on a packed record (legal moves only)."""
    rec = np.ascontiguousarray(rec, dtype=PACKED_DTYPE).reshape(1)
    out = np.zeros(1, dtype=PACKED_DTYPE)
    check(_lib.load().spx_pos_apply_uci(rec.ctypes.data, uci.encode(), out.ctypes.data))
    return out[0]


def random_positions(count, seed=1, min_ply=8, max_ply=120, dfrc_every=4):
    """Seeded random legal positions (random playouts), the synthetic batches of BASELINE config 2."""
    out = np.zeros(count, dtype=PACKED_DTYPE)
    check(_lib.load().spx_random_positions(seed, count, min_ply, max_ply, dfrc_every, out.ctypes.data))
    return out


def random_successors(positions, seed=1):
    """One uniformly random legal move per record -> (successor records, moved mask)."""
    pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
    out = np.zeros(pos.shape[0], dtype=PACKED_DTYPE)
    moved = np.zeros(pos.shape[0], dtype=np.uint8)
    check(_lib.load().spx_random_successors(seed, pos.ctypes.data, pos.shape[0], out.ctypes.data, moved.ctypes.data))
    return out, moved.astype(bool)


def viri_expand(data, with_filter=False):
    """viriformat game stream (bytes) -> (positions before each played move with eval/wdl filled, n_games[, unfiltered
    mask: the positions the reference's marlinformat output would keep])."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n, g = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.spx_viri_expand(buf.ctypes.data, buf.size, None, None, None, 0, ctypes.byref(n), ctypes.byref(g)))
    out = np.zeros(n.value, dtype=PACKED_DTYPE)
    keep = np.zeros(n.value, dtype=np.uint8) if with_filter else None
    check(lib.spx_viri_expand(buf.ctypes.data, buf.size, out.ctypes.data, None, None if keep is None else keep.ctypes.data,
                              n.value, ctypes.byref(n), ctypes.byref(g)))
    return (out, g.value, keep.astype(bool)) if with_filter else (out, g.value)


def viri_to_marlinformat(data):
    """viriformat stream -> the bytes datagen's marlinformat output holds for the same games (spx_viri_to_marlinformat):
    unfiltered positions as PackedBoard records, eval = recorded score, wdl = outcome. -> (records, n_games)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n, g = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.spx_viri_to_marlinformat(buf.ctypes.data, buf.size, None, 0, ctypes.byref(n), ctypes.byref(g)))
    out = np.zeros(n.value, dtype=PACKED_DTYPE)
    check(lib.spx_viri_to_marlinformat(buf.ctypes.data, buf.size, out.ctypes.data, n.value, ctypes.byref(n), ctypes.byref(g)))
    return out, g.value


def viri_to_fen(data):
    """viriformat stream -> the text datagen's "fen" output holds for the same games (spx_viri_to_fen):
    "<fen> | <score> | <0.0 / 0.5 / 1.0>" per unfiltered position. -> (text, n_games)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    n, g = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.spx_viri_to_fen(buf.ctypes.data, buf.size, None, 0, ctypes.byref(n), ctypes.byref(g)))
    out = ctypes.create_string_buffer(max(1, n.value))
    check(lib.spx_viri_to_fen(buf.ctypes.data, buf.size, out, n.value, ctypes.byref(n), ctypes.byref(g)))
    return out.raw[: n.value].decode(), g.value


def viri_random_game(seed, plies=80, dfrc=False):
    lib = _lib.load()
    buf = np.zeros(32 + 4 * (plies + 1), dtype=np.uint8)
    n = ctypes.c_size_t()
    check(lib.spx_viri_random_game(seed, plies, int(dfrc), buf.ctypes.data, buf.size, ctypes.byref(n)))
    return buf[: n.value].tobytes()


def debug_features(rec, colour):
    lib = _lib.load()
    rec = np.ascontiguousarray(rec, dtype=PACKED_DTYPE).reshape(1)
    psq = np.empty(32, dtype=np.uint32)
    thr = np.empty(256, dtype=np.uint32)
    n1, n2 = ctypes.c_int(), ctypes.c_int()
    check(lib.spx_debug_features(rec.ctypes.data, colour, psq.ctypes.data, ctypes.byref(n1), thr.ctypes.data, ctypes.byref(n2)))
    return psq[: n1.value].copy(), thr[: n2.value].copy()


def debug_delta(parent, child, colour):
    """Host emulation of the update kernel's delta derivation -> dict(psq_sub, psq_add, thr_sub, thr_add, refresh)."""
    lib = _lib.load()
    a = np.ascontiguousarray(parent, dtype=PACKED_DTYPE).reshape(1)
    b = np.ascontiguousarray(child, dtype=PACKED_DTYPE).reshape(1)
    bufs = [np.empty(n, dtype=np.uint32) for n in (8, 8, 288, 288)]
    counts = [ctypes.c_int() for _ in range(4)]
    refresh = ctypes.c_int()
    args = []
    for buf, cnt in zip(bufs, counts):
        args += [buf.ctypes.data, ctypes.byref(cnt)]
    check(lib.spx_debug_delta(a.ctypes.data, b.ctypes.data, colour, *args, ctypes.byref(refresh)))
    out = {k: buf[: cnt.value].copy() for k, buf, cnt in zip(("psq_sub", "psq_add", "thr_sub", "thr_add"), bufs, counts)}
    out["refresh"] = bool(refresh.value)
    return out


def count_rows(positions):
    """(psq_rows, threat_rows) gathered by a full refresh of the batch, both perspectives summed."""
    pos = np.ascontiguousarray(positions, dtype=PACKED_DTYPE)
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    check(_lib.load().spx_count_rows(pos.ctypes.data, pos.shape[0], ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def legal_moves(rec):
    """Host chess core: (viriformat move words, child records, in_check) of one packed record."""
    rec = np.ascontiguousarray(rec, dtype=PACKED_DTYPE).reshape(1)
    moves = np.zeros(256, dtype=np.uint16)
    children = np.zeros(256, dtype=PACKED_DTYPE)
    n, chk = ctypes.c_int(), ctypes.c_int()
    check(_lib.load().spx_pos_legal_moves(rec.ctypes.data, moves.ctypes.data, children.ctypes.data, ctypes.byref(n),
                                          ctypes.byref(chk)))
    return moves[: n.value].copy(), children[: n.value].copy(), bool(chk.value)


def perft(fen, depth):
    return int(_lib.load().spx_perft(fen.encode(), depth))
