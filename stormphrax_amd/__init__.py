"""stormphrax_amd: MI355X-native batched NNUE evaluator for Stormphrax (hot path only).

The product is the C-ABI shared library `libspx_nnue.so` (include/spx_nnue.h; sources in stormphrax_amd/csrc,
hand-written HIP for gfx950). This package is the Python plumbing the tests and bench.py use.
"""
from .nnue import (  # noqa: F401
    ADJUST_EVAL,
    ADJUST_STATIC,
    ADJUST_WDL,
    ADJUST_WHITE_POV,
    PACKED_DTYPE,
    DeviceGroup,
    Network,
    NnueState,
    adjust_params,
    apply_uci,
    count_rows,
    debug_delta,
    debug_features,
    device_count,
    legal_moves,
    perft,
    position_to_fen,
    positions_from_fens,
    positions_to_mailboxes,
    random_positions,
    random_successors,
    synthetic_net_bytes,
    viri_expand,
    viri_to_fen,
    viri_to_marlinformat,
    viri_random_game,
)
