"""The PRODUCT library - stormphrax_amd/libspx_nnue.so, what an engine links: exactly the functions of include/spx_nnue.h - evaluated
directly, with ctypes prototypes written out here from the header (the harness of the other GPU tests loads libspx_nnue_dev.so: the
same objects plus the test entry points). Reference behaviour under test: NnueState::evaluateOnce (src/eval/nnue_state.cpp:612-634)
against the compiled reference's own values (tests/golden/evals.jsonl)."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_P = ctypes.c_void_p


@pytest.fixture(scope="module")
def product():
    from stormphrax_amd import _lib

    lib = ctypes.CDLL(_lib.PRODUCT_PATH)
    lib.spx_last_error.restype = ctypes.c_char_p
    lib.spx_net_load.argtypes = [_P, ctypes.c_size_t, ctypes.POINTER(_P)]
    lib.spx_net_free.argtypes = [_P]
    lib.spx_net_free.restype = None
    lib.spx_ctx_create_opts.argtypes = [_P, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p, ctypes.POINTER(_P)]
    lib.spx_ctx_set_option.argtypes = [_P, ctypes.c_char_p, ctypes.c_int64]
    lib.spx_ctx_sliced_ft.argtypes = [_P, ctypes.c_size_t]
    lib.spx_ctx_destroy.argtypes = [_P]
    lib.spx_ctx_destroy.restype = None
    lib.spx_eval_full.argtypes = [_P, _P, ctypes.c_size_t, _P]
    assert not hasattr(lib, "spx_debug_enable_test_hooks") and not hasattr(lib, "spx_synth_net")
    return lib


def _golden_positions(sp):
    recs = [json.loads(line) for line in open(os.path.join(os.path.dirname(__file__), "golden", "evals.jsonl"))]
    return recs, sp.positions_from_fens([r["fen"] for r in recs])


@pytest.mark.parametrize("preset", ["tame", "realistic"])
def test_product_library_evaluates_the_reference_goldens(sp, product, net_blob, preset):
    """2 119 positions x the compiled reference's evaluateOnce values through libspx_nnue.so itself: the one-kernel path (a default
    context: 2 119 < 16 384) and the column-sliced pipeline (options given to spx_ctx_create_opts - the thread-safe replacement of
    round 5's environment rewrite)."""
    recs, pos = _golden_positions(sp)
    want = np.array([r[preset] for r in recs], dtype=np.int32)
    blob = np.ascontiguousarray(net_blob(preset))
    net = _P()
    assert product.spx_net_load(blob.ctypes.data, blob.size, ctypes.byref(net)) == 0, product.spx_last_error()
    try:
        for options, sliced in ((None, False), (b"ftx_min=1024,tiny_batch_max=0", True)):
            ctx = _P()
            assert product.spx_ctx_create_opts(net, 0, 4096, 0, options, ctypes.byref(ctx)) == 0, product.spx_last_error()
            try:
                assert bool(product.spx_ctx_sliced_ft(ctx, len(pos)) & 1) == sliced
                got = np.full(len(pos), -1, dtype=np.int32)
                assert product.spx_eval_full(ctx, pos.ctypes.data, len(pos), got.ctypes.data) == 0, product.spx_last_error()
                bad = np.nonzero(got != want)[0]
                assert bad.size == 0, (options, recs[bad[0]]["fen"], int(got[bad[0]]), int(want[bad[0]]))
            finally:
                product.spx_ctx_destroy(ctx)
    finally:
        product.spx_net_free(net)


def test_product_library_knows_no_fault_injection_hooks(sp, product, net_blob):
    """ftx_fail_after / ftx_fail_launch exist behind spx_debug_enable_test_hooks, which the product library does not
    export - there they are unknown options, at creation and afterwards; a malformed option string is refused too."""
    blob = np.ascontiguousarray(net_blob("tame"))
    net = _P()
    assert product.spx_net_load(blob.ctypes.data, blob.size, ctypes.byref(net)) == 0
    try:
        ctx = _P()
        for text in (b"ftx_fail_after=0", b"ftx_fail_launch=1", b"no_such_option=1", b"ftx_min", b"ftx_min=abc"):
            assert product.spx_ctx_create_opts(net, 0, 1024, 0, text, ctypes.byref(ctx)) == 1 and not ctx.value, text
        assert product.spx_ctx_create_opts(net, 0, 1024, 0, b"scratch_cap=2048,compact_rows=0", ctypes.byref(ctx)) == 0
        try:
            assert product.spx_ctx_set_option(ctx, b"ftx_fail_launch", 0) == 1
            assert b"unknown option" in product.spx_last_error()
            assert product.spx_ctx_set_option(ctx, b"ftx_auto_calibrate", 0) == 0
        finally:
            product.spx_ctx_destroy(ctx)
    finally:
        product.spx_net_free(net)


def test_a_call_under_stream_capture_does_not_calibrate():
    """the first big batch of a context chooses the gather's hot rows and the host waits for that inside the call - which
    a stream that is being captured into a hipGraph does not allow. Such a call runs with the set as it is (here: empty), the graph
    replays to the right scores, and the first call outside a capture calibrates. tests/_capture_worker.py, a process of its own: the
    capture is driven through torch, which must initialise its HIP runtime before the library is loaded (this pytest process did it
    the other way round, as in test_config4_hbm_filling_batch)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "_capture_worker.py")], cwd=root, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "capture ok" in out.stdout
