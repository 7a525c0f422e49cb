"""Worker for tests/test_distributed_cpu.py: world_size-2 gloo run of the sharding / gather / reduction plumbing that
bench.py uses with nccl (RCCL). The per-shard evaluator here is the CPU oracle (there is no GPU in this container)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import stormphrax_amd as sp  # noqa: E402
from conftest import Oracle  # noqa: E402
from stormphrax_amd.distributed import Group, shard_bounds  # noqa: E402


def main():
    n = int(sys.argv[1])
    out_path = sys.argv[2]
    group = Group(backend="gloo")
    positions = sp.random_positions(n, seed=77)          # every rank derives the same global batch...
    lo, hi = shard_bounds(n, group.rank, group.world)     # ...and evaluates only its contiguous shard
    oracle = Oracle()
    # the net image travels from rank 0 to the others (bench.py: RCCL broadcast at start-up), as SURVEY 8e lists
    blob = group.broadcast_bytes(sp.synthetic_net_bytes("tame") if group.rank == 0 else None)
    assert np.array_equal(blob, sp.synthetic_net_bytes("tame")), "broadcast net image differs from rank 0's"
    oracle.use(blob, "tame")
    mail, stm = sp.positions_to_mailboxes(positions[lo:hi])
    local = oracle.eval_mailboxes(mail, stm)
    full = group.gather_scores(local, n)
    checksum = group.sum_int(int(local.astype(np.int64).sum()))
    slowest = group.max_float(1.0 + group.rank)
    group.barrier()
    if group.rank == 0:
        with open(out_path, "w") as f:
            json.dump({"scores": full.tolist(), "checksum": checksum, "slowest": slowest, "world": group.world}, f)
    group.close()


if __name__ == "__main__":
    main()
