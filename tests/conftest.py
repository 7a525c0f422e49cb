"""Shared fixtures. `-m "not gpu"` tests run on CPU (oracle, host logic, ABI surface); `-m gpu` tests are the parity
tests proper and call the HIP path through the C ABI.

The oracle (oracle/spx_oracle.c, plain C) is TEST INFRASTRUCTURE: it is loaded here and nowhere in the product.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


class Oracle:
    """ctypes view of oracle/libspx_oracle.so (built on demand with gcc)."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "libspx_oracle.so")
        src = os.path.join(ROOT, "oracle", "spx_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
        self.lib = ctypes.CDLL(so)
        self.lib.spxo_init.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        self.lib.spxo_eval_mailboxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        self.lib.spxo_eval_fen.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32)]
        self.lib.spxo_features_mailbox.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
            ctypes.POINTER(ctypes.c_int)]
        self.lib.spxo_eval_accumulators.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.lib.spxo_eval_accumulators.restype = ctypes.c_int32
        self.lib.spxo_adjust.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                         ctypes.c_int, ctypes.c_int32, ctypes.c_int32]
        self.lib.spxo_adjust.restype = ctypes.c_int32
        self._preset = None

    def adjust(self, mailbox, stm, halfmove, raw, contempt=(0, 0), optimism=(0, 0), stages=3, correction=None):
        """adjustStatic / adjustEval restatement with the reference's default tunables (tunable.h:161-169)."""
        mailbox = np.ascontiguousarray(mailbox, dtype=np.uint8)
        params = np.array([*contempt, *optimism, 48, 442, 461, 637, 1223, 26000, 2024, 1005], dtype=np.int32)
        return int(self.lib.spxo_adjust(mailbox.ctypes.data, int(stm), int(halfmove), params.ctypes.data, stages,
                                        0 if correction is None else 1, 0 if correction is None else int(correction),
                                        int(raw)))

    def use(self, blob, tag):
        if self._preset != tag:
            blob = np.ascontiguousarray(blob)
            assert self.lib.spxo_init(blob.ctypes.data, blob.size) == 0
            self._preset = tag

    def eval_fen(self, fen):
        out = ctypes.c_int32()
        assert self.lib.spxo_eval_fen(fen.encode(), ctypes.byref(out)) == 0, fen
        return out.value

    def eval_mailboxes(self, mail, stm):
        mail = np.ascontiguousarray(mail, dtype=np.uint8)
        stm = np.ascontiguousarray(stm, dtype=np.uint8)
        out = np.empty(mail.shape[0], dtype=np.int32)
        assert self.lib.spxo_eval_mailboxes(mail.ctypes.data, stm.ctypes.data, mail.shape[0], out.ctypes.data) == 0
        return out

    def ft(self, mailbox, stm):
        mailbox = np.ascontiguousarray(mailbox, dtype=np.uint8)
        out = np.empty(1024, dtype=np.uint8)
        assert self.lib.spxo_ft_mailbox(mailbox.ctypes.data_as(ctypes.c_void_p), int(stm), out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out

    def features(self, mailbox, colour):
        mailbox = np.ascontiguousarray(mailbox, dtype=np.uint8)
        psq = np.empty(32, dtype=np.uint32)
        thr = np.empty(256, dtype=np.uint32)
        n1, n2 = ctypes.c_int(), ctypes.c_int()
        assert self.lib.spxo_features_mailbox(mailbox.ctypes.data, colour, psq.ctypes.data, ctypes.byref(n1),
                                              thr.ctypes.data, ctypes.byref(n2)) == 0
        return psq[: n1.value].copy(), thr[: n2.value].copy()


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def sp():
    import stormphrax_amd

    return stormphrax_amd


_NETS = {}


MIXED_WIDE_ROWS = np.arange(11264) % 3 == 0


def _mixed_rows_net(sp):
    """The wild synthetic net with every third piece-square row pushed out of the i8 range (values 128, -129, +-3000,
    i16 extremes in one column) and the i8 boundary values -128 / 127 planted in the others: exercises the per-row
    choice between the 1 KiB u8 copy and the 2 KiB i16 row of the feature-transformer kernels."""
    blob = np.array(sp.synthetic_net_bytes("wild"), copy=True)
    psq = blob[64 : 64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024)
    rng = np.random.default_rng(9)
    cols = rng.integers(0, 1024, size=11264)
    vals = rng.choice(np.array([128, -129, 3000, -3000, 32767, -32768], dtype=np.int16), size=11264)
    wide = np.nonzero(MIXED_WIDE_ROWS)[0]
    psq[wide, cols[wide]] = vals[wide]
    narrow = np.nonzero(~MIXED_WIDE_ROWS)[0]
    psq[narrow, cols[narrow]] = rng.choice(np.array([127, -128], dtype=np.int16), size=narrow.size)
    return blob


NEAR_ROW_KIND = np.arange(11264) % 4  # 0: compact, 1: 1-31 wide weights, 2: 33-60 (stays wide), 3: exactly 32


def _near_rows_net(sp):
    """The wild synthetic net with wide weights sprinkled into three quarters of the piece-square rows: 1-31 and exactly 32
    per row (near-compact: 1 KiB copy + remainders in the full-refresh kernel) and 33-60 (wide rows). Values cover both
    signs, the first values outside i8 and the i16 extremes; several land in the same 16-column lane group."""
    blob = np.array(sp.synthetic_net_bytes("wild"), copy=True)
    psq = blob[64 : 64 + 11264 * 1024 * 2].view("<i2").reshape(11264, 1024)
    rng = np.random.default_rng(16)
    pool = np.array([128, -129, 255, -256, 1000, -1000, 3000, -3000, 32767, -32768, 200, -200], dtype=np.int16)
    for r in range(11264):
        kind = NEAR_ROW_KIND[r]
        if kind == 0:
            continue
        n = int(rng.integers(1, 32)) if kind == 1 else (int(rng.integers(33, 61)) if kind == 2 else 32)
        cols = rng.choice(1024, size=n, replace=False)
        if n >= 4:  # neighbours: one lane owns several remainders, columns on both sides of 512
            cols[:4] = [(cols[0] // 8) * 8 + k for k in (0, 1, 6, 7)] if cols[0] % 1024 < 1016 else cols[:4]
            cols = np.unique(cols)
            while cols.size < n:
                cols = np.unique(np.append(cols, rng.integers(0, 1024)))
        psq[r, cols] = rng.choice(pool, size=cols.size)
    return blob


@pytest.fixture(scope="session")
def net_blob(sp):
    def get(preset):
        if preset not in _NETS:
            _NETS[preset] = (_mixed_rows_net(sp) if preset == "mixed" else _near_rows_net(sp) if preset == "near"
                             else sp.synthetic_net_bytes(preset))
        return _NETS[preset]

    return get
