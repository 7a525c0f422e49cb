"""Multi-GPU plumbing: one process per GPU, positions sharded contiguously, NO collective on the data path.

Positions (and games) are independent, so the only communication is (a) rendezvous, (b) an optional all_gather of the
4-byte scores back to every rank, (c) a MAX all_reduce of the timed region and a SUM all_reduce of a checksum for the
report. `backend="nccl"` is RCCL over xGMI on the GPU box; the same code runs with `gloo` on CPU in
tests/test_distributed_cpu.py.
"""
import os

import numpy as np

# The host driver of these boxes only supports dmabuf IPC: without this RCCL (and any sharing of device memory between the
# ranks of one node) fails with "hipIpcGetMemHandle: invalid argument". It must be in the environment before the HIP runtime
# starts, so the entry scripts import this module before torch.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Group:
    """Thin wrapper over torch.distributed that degrades to a no-op for world size 1 (SPX_FORCE_DIST=1 initialises the
    process group even then: a single-GPU box can put every collective below through the real RCCL once)."""

    def __init__(self, backend="nccl", device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1 or os.environ.get("SPX_FORCE_DIST") == "1":
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kwargs)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def _tensor(self, values, dtype):
        import torch

        t = torch.tensor(values, dtype=dtype)
        return t.to(self.device) if self.device is not None else t

    def max_float(self, x):
        if not self.dist:
            return float(x)
        import torch

        t = self._tensor([x], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_floats(self, x):
        """Every rank's value of `x`, in rank order, on every rank (per-rank timings for the report: a straggler shows)."""
        if not self.dist:
            return [float(x)]
        import torch

        mine = self._tensor([float(x)], torch.float64)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [float(p.item()) for p in parts]

    def backend_world(self):
        """(backend name, world size) as torch.distributed reports them - what actually carries the collectives."""
        if not self.dist:
            return "none", 1
        return str(self.dist.get_backend()), int(self.dist.get_world_size())

    def sum_int(self, x):
        if not self.dist:
            return int(x)
        import torch

        t = self._tensor([int(x)], torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def broadcast_bytes(self, data, src=0):
        """Rank `src`'s uint8 array on every rank (the 89 MB net image at start-up: SURVEY 8e's ncclBroadcast of the
        weight blob; over RCCL the payload travels GPU to GPU across xGMI). Ranks other than `src` pass None."""
        if not self.dist:
            return np.ascontiguousarray(data, dtype=np.uint8)
        import torch

        size = self._tensor([0 if data is None else int(np.asarray(data).size)], torch.int64)
        self.dist.broadcast(size, src=src)
        n = int(size.item())
        if self.rank == src:
            t = torch.from_numpy(np.ascontiguousarray(data, dtype=np.uint8).copy())
        else:
            t = torch.empty(n, dtype=torch.uint8)
        if self.device is not None:
            t = t.to(self.device)
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def gather_scores(self, local_scores, n_total):
        """all_gather of per-rank int32 score shards (contiguous sharding) -> full array on every rank."""
        local = np.ascontiguousarray(local_scores, dtype=np.int32)
        if not self.dist:
            return local
        import torch

        sizes = [shard_bounds(n_total, r, self.world) for r in range(self.world)]
        width = max(hi - lo for lo, hi in sizes)
        pad = np.zeros(width, dtype=np.int32)
        pad[: local.size] = local
        mine = torch.from_numpy(pad)
        if self.device is not None:
            mine = mine.to(self.device)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        out = np.empty(n_total, dtype=np.int32)
        for (lo, hi), part in zip(sizes, parts):
            out[lo:hi] = part.cpu().numpy()[: hi - lo]
        return out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
