#!/usr/bin/env python3
"""Batched self-play on the GPU (BASELINE config 4 shape): N concurrent games per GPU, every candidate move of every game
evaluated in one incremental update+eval batch per ply (depth-1 policy), viriformat output.

    python tools/spx_selfplay.py --games 4096 --target 8192 --out /tmp/games            # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/spx_selfplay.py --games 4096 ...

Games are independent (config 4: "games sharded 2/4/8 MI355X via RCCL/xGMI"): each rank plays its own games on its own
GPU (seed + rank) and writes <out>.<rank>.vf; the communication - over RCCL (torch.distributed backend "nccl") - is the
broadcast of rank 0's net image at start-up and the final SUM / MAX of the counters. Prints one JSON line on rank 0.
SPX_BENCH_BACKEND=gloo + SPX_BENCH_SHARE_GPU=1 run the same control flow with every rank on GPU 0 (single-GPU boxes)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks (before the HIP runtime starts)
import stormphrax_amd as sp  # noqa: E402
from stormphrax_amd.distributed import Group  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--target", type=int, default=8192, help="games to finish per GPU")
    ap.add_argument("--max-plies", type=int, default=300)
    ap.add_argument("--temperature", type=int, default=30)
    ap.add_argument("--dfrc", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--host-movegen", action="store_true", help="generate moves with the host chess core instead of on the GPU")
    ap.add_argument("--search-nodes", type=int, default=0,
                    help="live fixed-node search: expansions per search (SPX_SELFPLAY_SEARCH_NODES; 0 = the depth-1 policy)")
    ap.add_argument("--preset", default="tame")
    ap.add_argument("--net")
    ap.add_argument("--out")
    ap.add_argument("--format", default="viriformat", choices=["viriformat", "marlinformat", "fen"],
                    help="datagen's output formats (datagen.cpp:340-346). The games are always recorded as viriformat "
                         "(<out>.<rank>.vf); marlinformat / fen convert that file afterwards into what the reference would have "
                         "written for the same games (<out>.<rank>.bin / .txt: unfiltered positions only) and remove it")
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    import torch

    from stormphrax_amd.distributed import env_rank

    # SPX_BENCH_SHARE_GPU=1 (debug, as in bench.py): every rank on GPU 0, to exercise the N > 1 control flow on one GPU
    device = 0 if os.environ.get("SPX_BENCH_SHARE_GPU") == "1" else env_rank()[1] % max(1, torch.cuda.device_count())
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")  # "nccl" is RCCL over xGMI on ROCm
    if backend == "nccl":
        torch.cuda.set_device(device)
    group = Group(backend=backend, device=torch.device("cuda", device) if backend == "nccl" else None)
    blob = None
    if group.rank == 0:
        blob = np.fromfile(args.net, dtype=np.uint8) if args.net else sp.synthetic_net_bytes(args.preset)
    net = sp.Network(group.broadcast_bytes(blob))
    state = sp.NnueState(net, device=device, max_batch=args.games * 64)
    out = f"{args.out}.{group.rank}.vf" if args.out else None
    stats = state.selfplay(args.games, args.target, out_path=out, max_plies=args.max_plies, dfrc=args.dfrc,
                           temperature_cp=args.temperature, seed=args.seed + group.rank, host_threads=args.threads,
                           host_movegen=args.host_movegen, search_nodes=args.search_nodes)
    written = None
    if out and args.format != "viriformat":
        data = open(out, "rb").read()
        if args.format == "marlinformat":
            records, _ = sp.viri_to_marlinformat(data)
            written = f"{args.out}.{group.rank}.bin"
            records.tofile(written)
        else:
            text, _ = sp.viri_to_fen(data)
            written = f"{args.out}.{group.rank}.txt"
            open(written, "w").write(text)
        os.remove(out)
    slowest = group.max_float(stats["seconds"])
    total = {k: group.sum_int(stats[k]) for k in ("games", "positions", "evals", "steps")}
    gpu_seconds = group.max_float(stats["gpu_seconds"])
    outcomes = [group.sum_int(v) for v in stats["outcomes"]]
    if group.rank == 0:
        print(json.dumps({
            "metric": "selfplay_leaf_evals_per_sec", "value": total["evals"] / slowest, "unit": "evals/s",
            "n_gpus": group.world, "collectives": backend, "net_digest": "%016x" % net.digest, "games_per_gpu": args.games, "games": total["games"], "positions": total["positions"],
            "positions_per_sec": total["positions"] / slowest, "games_per_sec": total["games"] / slowest,
            "seconds": slowest, "gpu_call_seconds": gpu_seconds, "gpu_call_fraction": gpu_seconds / slowest,
            "outcomes_white_loss_draw_win": outcomes, "host_threads": args.threads or "auto", "format": args.format,
            "move_generation": "host chess core" if args.host_movegen else "device (spx_movegen_kernel)",
            "policy": ("live fixed-node search, %d expansions (iterative-deepening alpha-beta, leaves = NNUE(child))" % args.search_nodes)
                      if args.search_nodes else
                      "depth-1: score(move) = -NNUE(child), uniform among moves within %d cp of the best" % args.temperature,
            "nodes_expanded": total["steps"] if args.search_nodes else None,
        }))
    group.close()


if __name__ == "__main__":
    main()
