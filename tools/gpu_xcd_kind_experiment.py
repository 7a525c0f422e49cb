#!/usr/bin/env python3
"""Experiment: would routing update records to XCDs by the KIND of the moved piece raise the L2 hit rate of the delta rows?
Every delta row of a move involves the moved piece, so an XCD that only sees, say, white-knight moves works on ~1/6 of the
threat table instead of all of it. Emulated without new kernels: 65 536 games move and unmake the same move (A -> B -> A ...),
the games are permuted on the host (as generated / sorted by moved-piece kind / by kind and arrival square), and the library
was built twice - the product's chunked round-robin dealing, and a variant whose ItemWalk::item() returned xcd * tEnd + t (XCD x
walks the x-th contiguous eighth of the records; not kept in the source). Result: profiles/r03_ab_update_records_routed_by_moved_piece.txt. Prints the update kernel (+ rebuild pass) time per ply for each combination (SPX_LIB selects the build)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import stormphrax_amd as sp
    from stormphrax_amd import _lib

    lib = _lib.load()
    G = 65536
    a = sp.random_positions(G, seed=777, min_ply=6, max_ply=60, dfrc_every=4)
    b, moved = sp.random_successors(a, seed=1000)
    ma, _ = sp.positions_to_mailboxes(a)
    mb, _ = sp.positions_to_mailboxes(b)
    diff = ma != mb
    # the arrival square: changed, occupied afterwards (castling: the lowest such square; good enough for a sort key)
    arrive = np.argmax(diff & (mb != mb.max()), axis=1)
    kind = mb[np.arange(G), arrive].astype(np.int64)
    orders = {"as generated": np.arange(G), "sorted by moved-piece kind": np.argsort(kind, kind="stable"),
              "sorted by kind and arrival square": np.lexsort((arrive, kind))}
    st = sp.NnueState(sp.Network(sp.synthetic_net_bytes("tame")), device=0, max_batch=G)
    st.reserve_slots(2 * G)
    slots = [torch.arange(G, dtype=torch.int32, device="cuda"), torch.arange(G, 2 * G, dtype=torch.int32, device="cuda")]
    out = torch.empty(G, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    tag = os.path.basename(os.environ.get("SPX_LIB", "product"))
    for name, o in orders.items():
        boards = [torch.from_numpy(np.ascontiguousarray(x[o]).view(np.uint8).reshape(-1, 32)).cuda() for x in (a, b)]
        _lib.check(lib.spx_acc_refresh_device(st._h, boards[0].data_ptr(), slots[0].data_ptr(), G, stream))
        torch.cuda.synchronize()

        def ply(k):
            _lib.check(lib.spx_acc_update_eval_device(st._h, slots[k & 1].data_ptr(), slots[(k + 1) & 1].data_ptr(),
                                                      boards[(k + 1) & 1].data_ptr(), G, out.data_ptr(), stream))
        for k in range(40):
            ply(k)
        torch.cuda.synchronize()
        st.profile_begin(400)
        for k in range(400):
            ply(k)
        torch.cuda.synchronize()
        _, upd_ms, mlp_ms, calls = st.profile_end()
        full = torch.empty(G, dtype=torch.int32, device="cuda")
        st.evaluate_once_device(boards[0].data_ptr(), G, full.data_ptr(), stream)
        torch.cuda.synchronize()
        print("%-14s %-36s update + rebuild %.1f us per ply, exact %s, kinds %d" % (tag, name, upd_ms / calls * 1e3, bool(torch.equal(full, out)), len(set(kind.tolist()))))


if __name__ == "__main__":
    main()
