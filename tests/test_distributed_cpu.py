"""N > 1 path on CPU: world_size 2 over gloo (the GPU box uses the same code over nccl/RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from stormphrax_amd.distributed import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharded_eval_matches_single_process(tmp_path, sp, oracle, net_blob):
    n = 301  # odd on purpose: ragged shards
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), str(n), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, capture_output=True)
    res = json.loads(out.read_text())
    positions = sp.random_positions(n, seed=77)
    mail, stm = sp.positions_to_mailboxes(positions)
    oracle.use(net_blob("tame"), "tame")
    want = oracle.eval_mailboxes(mail, stm)
    assert res["world"] == 2
    assert np.array_equal(np.array(res["scores"], dtype=np.int32), want)
    assert res["checksum"] == int(want.astype(np.int64).sum())
    assert res["slowest"] == 2.0


def test_bare_bench_command_spawns_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself instead of refusing.
    On this GPU-less container the ranks then stop at the "needs a GPU" check - the spawn is what is tested here; the
    full run is tests/test_gpu_configs.py::test_bare_bench_command_launches_its_own_ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert "launching 2 ranks" in out.stderr
    import torch

    if not torch.cuda.is_available():
        assert out.returncode != 0 and "bench.py needs a GPU" in out.stderr
        assert "launch with torch.distributed.run" not in out.stderr
