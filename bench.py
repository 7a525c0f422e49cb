#!/usr/bin/env python3
"""bench.py - batched NNUE position-evals/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one full-refresh pass of the hot path (feature extraction + FT accumulation + pairwise activation + int8
MFMA L1 + i32 tail) over ONE batch of synthetic positions already resident in HBM (BASELINE config 2: 65 536 seeded
random legal positions per GPU, synthetic net). N > 1: every rank evaluates its own shard (positions are independent -
no data-path collective), weak scaling; value = all ranks' positions / max-over-ranks time.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 200 --warmup 20

Rank 0 prints ONE JSON line. Extra objects: `roofline` (feature-transformer kernel, algorithmic gather bytes / HIP-event
kernel time vs the 8 TB/s HBM peak) and `cpu_baseline` (the compiled reference `oracle/_ref/sp_ref_probe_tame` when
present, else the C restatement, timed on a bounded sample of the same batch on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def cpu_baseline(sp, positions, blob, seconds):
    """Reference CPU path on a bounded sample of the batch (baseline only, never the target)."""
    sample = positions[:4096]
    probe = os.path.join(ROOT, "oracle", "_ref", "sp_ref_probe_tame")
    cores = os.cpu_count() or 1
    if os.path.exists(probe):
        try:
            cmds = "".join(f"add {sp.position_to_fen(p)}\n" for p in sample)
            cmds += f"bench {cores} {seconds}\nquit\n"
            out = subprocess.run([probe], input=cmds, capture_output=True, text=True, timeout=seconds + 120).stdout
            line = [ln for ln in out.splitlines() if ln.startswith("B ")][0].split()
            return {
                "value": float(line[1]), "unit": "evals/s", "cores": cores, "kind": "reference",
                "sample": f"compiled Stormphrax 8.0.2 (AVX2 build) NnueState::evaluateOnce looped over the first "
                          f"{len(sample)} positions of the batch for {seconds:.0f} s on {cores} threads",
            }
        except Exception as exc:  # fall through to the port
            print(f"[bench] reference probe failed ({exc}); timing the C restatement instead", file=sys.stderr)
    so = os.path.join(ROOT, "oracle", "libspx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
    oracle = ctypes.CDLL(so)
    oracle.spxo_init.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    oracle.spxo_eval_mailboxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert oracle.spxo_init(blob.ctypes.data, blob.size) == 0
    mail, stm = sp.positions_to_mailboxes(sample)
    out = np.empty(len(sample), dtype=np.int32)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        oracle.spxo_eval_mailboxes(mail.ctypes.data, stm.ctypes.data, len(sample), out.ctypes.data)
        done += len(sample)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"scalar C restatement (oracle/spx_oracle.c) over the first {len(sample)} positions, {dt:.1f} s, 1 thread"}


def pmc_traffic(args):
    """HBM/fabric bytes per launch of the FT kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/ft_traffic_pmc.json; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md, + WRITE_SIZE).
    bench.py cannot collect counters on itself; null when the run's configuration differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "ft_traffic_pmc.json")
    try:
        rec = json.load(open(path))
    except OSError:
        return None
    if rec.get("batch") != args.batch or rec.get("preset") != args.preset:
        return None
    return rec["traffic_bytes_per_launch"] / 1e9 / rec["ft_kernel_ms"] * 1e3  # GB/s, comparable with `achieved`


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="positions per GPU per step")
    ap.add_argument("--preset", default="tame", choices=["tame", "wild", "extreme"])
    ap.add_argument("--distinct", type=int, default=0,
                    help="diagnostic: tile this many distinct positions to fill the batch (cache-locality ablation)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import stormphrax_amd as sp

    from stormphrax_amd.distributed import Group, env_rank

    rank, local_rank, world = env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the evaluator has no CPU path)")
    torch.cuda.set_device(local_rank)
    group = Group(backend="nccl", device=torch.device("cuda", local_rank))  # RCCL over xGMI; no-op for 1 GPU

    # ---- workload: this rank's shard of seeded random legal positions, resident in HBM ----
    blob = sp.synthetic_net_bytes(args.preset)
    net = sp.Network(blob)
    state = sp.NnueState(net, device=local_rank, max_batch=args.batch)
    positions = sp.random_positions(args.batch, seed=20260927 + rank, min_ply=8, max_ply=120, dfrc_every=4)
    if args.distinct:
        positions = np.resize(positions[: args.distinct], args.batch)
    d_pos = torch.from_numpy(positions.view(np.uint8).reshape(-1, 32)).cuda()
    d_out = torch.empty(args.batch, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        state.evaluate_once_device(d_pos.data_ptr(), args.batch, d_out.data_ptr(), stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    state.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sort_ms, ft_ms, mlp_ms, calls = state.profile_end()

    elapsed = group.max_float(elapsed)  # slowest rank defines the step time
    checksum = group.sum_int(int(d_out.to(torch.int64).sum().item()))  # checksum of checksums over all shards

    if rank == 0:
        psq_rows, thr_rows = sp.count_rows(positions)
        algo_bytes = 2048 * psq_rows + 1024 * thr_rows + 36 * args.batch  # per launch
        ft_avg_s = ft_ms / max(calls, 1) / 1e3
        achieved = algo_bytes / ft_avg_s / 1e9
        value = world * args.batch * args.steps / elapsed
        line = {
            "metric": "nnue_position_evals_per_sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i16 accumulate / i8 MFMA L1 / i32 tail",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: full-refresh NNUE forward on 65536 seeded random legal positions "
                            "per GPU (random playouts 8-120 plies, every 4th game DFRC), bit-exact vs CPU",
                "batch_per_gpu": args.batch,
                "net": f"synthetic CBNF '{net.name}' (Stormphrax 8.0.2 shape: (704x16+64368)->1024)x2->(32x2->64->1)x8",
                "parallelism": f"positions sharded over {world} GPU(s), no collective on the data path",
                "checksum": checksum,
                "kernel_ms": {"sort": sort_ms / max(calls, 1), "ft": ft_ms / max(calls, 1), "mlp": mlp_ms / max(calls, 1)},
            },
            "roofline": {
                "kernel": "spx_ft_kernel",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args),
                "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_position": algo_bytes / args.batch,
            },
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sp, positions, blob, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    group.close()


if __name__ == "__main__":
    main()
