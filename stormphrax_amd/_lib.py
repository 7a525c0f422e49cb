"""ctypes binding of the C ABI (include/spx_nnue.h) and of the test / measurement entry points (include/spx_nnue_dev.h).

The shared library is the product; this module is plumbing for the Python harnesses (tests, bench.py, tools). Two builds of
the same objects sit in the tree (csrc/Makefile): libspx_nnue.so exports the boundary and nothing else, libspx_nnue_dev.so adds
the dev entry points the harnesses need (synthetic nets, random positions, spx_debug_*) - the one loaded here. The module
fails loudly when the in-tree library is missing - there is no Python or CPU fallback for the evaluation path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPX_LIB lets kernel experiments point the harness at an alternative build of the same ABI (A/B runs on the GPU box)
LIB_PATH = os.environ.get("SPX_LIB") or os.path.join(_HERE, "libspx_nnue_dev.so")
PRODUCT_PATH = os.path.join(_HERE, "libspx_nnue.so")  # the boundary alone (tests/test_abi.py, tests/test_gpu_native_host.py)


class PackedPos(ctypes.Structure):
    """spx_packed_pos: marlinformat PackedBoard (reference src/datagen/marlinformat.h:32-84), 32 bytes."""

    _pack_ = 1
    _fields_ = [
        ("occupancy", ctypes.c_uint64),
        ("pieces", ctypes.c_uint8 * 16),
        ("stm_ep", ctypes.c_uint8),
        ("halfmove", ctypes.c_uint8),
        ("fullmove", ctypes.c_uint16),
        ("eval", ctypes.c_int16),
        ("wdl", ctypes.c_uint8),
        ("extra", ctypes.c_uint8),
    ]


assert ctypes.sizeof(PackedPos) == 32


class ThreatDesc(ctypes.Structure):
    _fields_ = [("attacker", ctypes.c_uint8), ("attacker_sq", ctypes.c_uint8), ("attacked", ctypes.c_uint8),
                ("attacked_sq", ctypes.c_uint8)]


class MoveDelta(ctypes.Structure):
    """spx_move_delta: the reference's UpdateContext (src/eval/nnue_state.h:28-31)."""

    _fields_ = [
        ("n_sub", ctypes.c_uint8), ("n_add", ctypes.c_uint8),
        ("n_threats_added", ctypes.c_uint8), ("n_threats_removed", ctypes.c_uint8),
        ("sub_piece", ctypes.c_uint8 * 2), ("sub_sq", ctypes.c_uint8 * 2),
        ("add_piece", ctypes.c_uint8 * 2), ("add_sq", ctypes.c_uint8 * 2),
        ("psq_refresh", ctypes.c_uint8 * 2), ("threat_refresh", ctypes.c_uint8 * 2),
        ("kings", ctypes.c_uint8 * 2), ("reserved", ctypes.c_uint8 * 2),
        ("pawns_before", ctypes.c_uint64 * 2), ("pawns_after", ctypes.c_uint64 * 2),
        ("threats_added", ThreatDesc * 128), ("threats_removed", ThreatDesc * 128),
    ]


assert ctypes.sizeof(MoveDelta) == 1080

class AdjustParams(ctypes.Structure):
    """spx_adjust_params (include/spx_nnue.h): eval::Contempt / Optimism and the scaling tunables of eval.cpp:30-67."""
    _fields_ = [("contempt", ctypes.c_int32 * 2), ("optimism", ctypes.c_int32 * 2),
                ("scaling_value", ctypes.c_int32 * 5), ("material_scaling_base", ctypes.c_int32),
                ("optimism_base", ctypes.c_int32), ("optimism_material_scale", ctypes.c_int32),
                ("stages", ctypes.c_uint32)]


class SelfplayParams(ctypes.Structure):
    _fields_ = [("n_games", ctypes.c_uint32), ("target_games", ctypes.c_uint32), ("max_plies", ctypes.c_uint32),
                ("opening_plies", ctypes.c_uint32), ("dfrc", ctypes.c_uint32), ("temperature_cp", ctypes.c_int32),
                ("host_threads", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("seed", ctypes.c_uint64)]


class SelfplayStats(ctypes.Structure):
    _fields_ = [("games", ctypes.c_uint64), ("positions", ctypes.c_uint64), ("evals", ctypes.c_uint64),
                ("steps", ctypes.c_uint64), ("outcomes", ctypes.c_uint64 * 3), ("seconds", ctypes.c_double),
                ("gpu_seconds", ctypes.c_double)]


# every symbol include/spx_nnue.h declares: (restype, argtypes)
_P = ctypes.c_void_p
SYMBOLS = {
    "spx_last_error": (ctypes.c_char_p, []),
    "spx_net_load": (ctypes.c_int, [_P, ctypes.c_size_t, ctypes.POINTER(_P)]),
    "spx_net_free": (None, [_P]),
    "spx_net_name": (ctypes.c_char_p, [_P]),
    "spx_synth_net_bytes": (ctypes.c_size_t, []),
    "spx_synth_net": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int, _P, ctypes.c_size_t]),
    "spx_fnv1a64": (ctypes.c_uint64, [_P, ctypes.c_size_t]),
    "spx_ctx_create": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(_P)]),
    "spx_ctx_create_ex": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(_P)]),
    "spx_ctx_create_opts": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p, ctypes.POINTER(_P)]),
    "spx_ctx_scratch_batch": (ctypes.c_size_t, [_P]),
    "spx_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "spx_group_create": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(_P)]),
    "spx_group_destroy": (None, [_P]),
    "spx_group_size": (ctypes.c_size_t, [_P]),
    "spx_group_member": (_P, [_P, ctypes.c_size_t]),
    "spx_group_shard": (ctypes.c_int, [_P, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_group_eval_full": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "spx_group_adjust": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, _P]),
    "spx_ctx_destroy": (None, [_P]),
    "spx_eval_full": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "spx_eval_full_device": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P]),
    "spx_ctx_sliced_ft": (ctypes.c_int, [_P, ctypes.c_size_t]),
    "spx_profile_begin": (ctypes.c_int, [_P, ctypes.c_size_t]),
    "spx_profile_last_prepare_ms": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double)]),
    "spx_profile_end": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_acc_reserve": (ctypes.c_int, [_P, ctypes.c_size_t]),
    "spx_acc_refresh": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t]),
    "spx_acc_update": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t]),
    "spx_acc_eval": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "spx_acc_refresh_device": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, _P]),
    "spx_acc_update_device": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, _P]),
    "spx_acc_update_eval": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, _P]),
    "spx_acc_update_chain_eval": (ctypes.c_int, [_P, ctypes.c_uint32, _P, _P, ctypes.c_size_t, _P]),
    "spx_acc_update_eval_device": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "spx_acc_update_eval_device_async": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_size_t, _P, ctypes.POINTER(ctypes.c_void_p)]),
    "spx_acc_replay_tree": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, _P, ctypes.c_size_t, _P, ctypes.POINTER(ctypes.c_double)]),
    "spx_acc_eval_device": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P]),
    "spx_net_digest": (ctypes.c_uint64, [_P]),
    "spx_net_psq_row_classes": (ctypes.c_int, [_P] + [ctypes.POINTER(ctypes.c_uint32)] * 3),
    "spx_ctx_compact_psq_rows": (ctypes.c_uint32, [_P]),
    "spx_ctx_near_psq_rows": (ctypes.c_uint32, [_P]),
    "spx_acc_update_eval_device_counted": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "spx_eval_full_device_async": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, ctypes.POINTER(ctypes.c_void_p)]),
    "spx_ctx_synchronize": (ctypes.c_int, [_P]),
    "spx_viri_expand_gpu": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                           ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_movegen": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "spx_movegen_device": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "spx_pos_legal_moves": (ctypes.c_int, [_P, _P, _P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "spx_host_alloc": (_P, [ctypes.c_size_t]),
    "spx_host_free": (None, [_P]),
    "spx_adjust_defaults": (None, [_P]),
    "spx_adjust": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, _P]),
    "spx_adjust_device": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, _P, _P, _P]),
    "spx_count_rows": (ctypes.c_int, [_P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "spx_ctx_count_rows": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "spx_ctx_set_option": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int64]),
    "spx_ctx_calibrate": (ctypes.c_int, [_P, _P, ctypes.c_size_t]),
    "spx_ctx_set_hot_rows": (ctypes.c_int, [_P, _P, ctypes.c_size_t]),
    "spx_ctx_get_hot_rows": (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "spx_debug_copy_ft": (ctypes.c_int, [_P, ctypes.c_size_t, _P]),
    "spx_debug_enable_test_hooks": (ctypes.c_int, [ctypes.c_int]),
    "spx_debug_ftx_block_times": (ctypes.c_int, [_P, ctypes.c_int, _P]),
    "spx_debug_ftx_plan": (ctypes.c_int, [_P, ctypes.c_int, _P]),
    "spx_debug_ftx_walk": (ctypes.c_int, [_P, ctypes.c_int, _P]),
    "spx_debug_ftx_lists": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_size_t, _P, _P]),
    "spx_pos_from_fen": (ctypes.c_int, [ctypes.c_char_p, _P]),
    "spx_pos_to_fen": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_size_t]),
    "spx_pos_to_mailbox": (ctypes.c_int, [_P, _P, ctypes.POINTER(ctypes.c_int)]),
    "spx_pos_apply_uci": (ctypes.c_int, [_P, ctypes.c_char_p, _P]),
    "spx_tree_expand_uci": (ctypes.c_int, [_P, ctypes.c_size_t, _P, _P, ctypes.c_size_t, _P]),
    "spx_pos_apply_uci_observed": (ctypes.c_int, [_P, ctypes.c_char_p, _P, _P]),
    "spx_acc_update_observed": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "spx_acc_update_observed_device": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "spx_random_positions": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "spx_random_positions_gpu": (ctypes.c_int, [_P, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "spx_random_successors": (ctypes.c_int, [ctypes.c_uint64, _P, ctypes.c_size_t, _P, _P]),
    "spx_viri_to_marlinformat": (ctypes.c_int, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_viri_to_fen": (ctypes.c_int, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_viri_expand": (ctypes.c_int, [_P, ctypes.c_size_t, _P, _P, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "spx_viri_random_game": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _P, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "spx_selfplay_run": (ctypes.c_int, [_P, _P, ctypes.c_char_p, _P]),
    "spx_group_selfplay_run": (ctypes.c_int, [_P, _P, ctypes.c_char_p, _P]),
    "spx_perft": (ctypes.c_uint64, [ctypes.c_char_p, ctypes.c_int]),
    "spx_debug_delta": (ctypes.c_int, [_P, _P, ctypes.c_int] + [_P, ctypes.POINTER(ctypes.c_int)] * 4 + [ctypes.POINTER(ctypes.c_int)]),
    "spx_debug_wdl": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "spx_debug_datagen_rules": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), _P, ctypes.POINTER(ctypes.c_int)]),
    "spx_debug_features": (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.POINTER(ctypes.c_int), _P, ctypes.POINTER(ctypes.c_int)]),
}

_lib = None


def load():
    """Load libspx_nnue_dev.so and declare prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback implementation."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class SpxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"spx error {code}: {message}")
        self.code = code


def check(rc):
    if rc != 0:
        raise SpxError(rc, load().spx_last_error().decode(errors="replace"))
