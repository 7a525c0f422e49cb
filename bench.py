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


def usable_cpus():
    """Host threads this process can actually run at once: logical CPUs, capped by the affinity mask and by the
    container's CPU quota (cgroup v2 cpu.max / v1 cfs quota). The MI355X boxes expose 256 logical CPUs with a quota of
    16: oversubscribing the quota makes the CPU baseline SLOWER (measured: 2.5e6 evals/s at 16 threads, 1.2e6 at 256)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sp, positions, blob, seconds):
    """Reference CPU path on a bounded sample of the batch (baseline only, never the target)."""
    sample = positions[:4096]
    probe, flavour = os.path.join(ROOT, "oracle", "_ref", "sp_ref_probe_tame"), "AVX2-BMI2 build"
    try:  # the reference's fastest flavour (build.mk:36 "avx512": VNNI + VBMI2) when this host can run it
        flags = set(next(ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")).split())
        fast = probe + "_avx512"
        if {"avx512f", "avx512bw", "avx512vl", "avx512_vnni", "avx512_vbmi2", "avx512vbmi"} <= flags and os.path.exists(fast):
            probe, flavour = fast, "AVX-512 VNNI/VBMI2 build"
    except (OSError, StopIteration):
        pass
    cores = usable_cpus()
    if os.path.exists(probe):
        try:
            cmds = "".join(f"add {sp.position_to_fen(p)}\n" for p in sample)
            cmds += f"bench {cores} {seconds}\nquit\n"
            out = subprocess.run([probe], input=cmds, capture_output=True, text=True, timeout=seconds + 120).stdout
            line = [ln for ln in out.splitlines() if ln.startswith("B ")][0].split()
            return {
                "value": float(line[1]), "unit": "evals/s", "cores": cores, "kind": "reference",
                "sample": f"compiled Stormphrax 8.0.2 ({flavour}) NnueState::evaluateOnce looped over the first "
                          f"{len(sample)} positions of the batch for {seconds:.0f} s on {cores} threads "
                          f"(= usable CPUs: {os.cpu_count()} logical, capped by affinity / cgroup quota)",
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


def delta_rows_sample(sp, parents, children, sample=256):
    """Mean (psq, threat) rows added+removed per position update (both perspectives), from exact set differences of the
    host-emulated feature lists on a sample - the algorithmic bytes of the incremental kernel."""
    psq = thr = 0
    n = min(sample, len(parents))
    for i in range(n):
        for c in (0, 1):
            p0, t0 = sp.debug_features(parents[i], c)
            p1, t1 = sp.debug_features(children[i], c)
            psq += len(set(p0.tolist()) ^ set(p1.tolist()))
            thr += len(set(t0.tolist()) ^ set(t1.tolist()))
    return psq / n, thr / n


def incremental_bench(args, sp, torch, group, rank, local_rank, world):
    """--mode incremental: G = --batch concurrent games per GPU; a step = one ply for every game (one spx_acc_update
    batch of G independent parent->child records + one spx_acc_eval batch). Boards come from seeded random playouts
    (32-ply chains walked forward then backward: an unmake is a one-move delta too)."""
    from stormphrax_amd import _lib

    lib = _lib.load()
    G, L = args.batch, 32
    blob = sp.synthetic_net_bytes(args.preset)
    net = sp.Network(blob)
    state = sp.NnueState(net, device=local_rank, max_batch=G)
    chain = [sp.random_positions(G, seed=777 + rank, min_ply=6, max_ply=60, dfrc_every=4)]
    for ply in range(L - 1):
        nxt, _ = sp.random_successors(chain[-1], seed=1000 + ply + 97 * rank)
        chain.append(nxt)
    d_boards = [torch.from_numpy(c.view(np.uint8).reshape(-1, 32)).cuda() for c in chain]
    state.reserve_slots(2 * G)
    slots = [torch.arange(G, dtype=torch.int32, device="cuda"), torch.arange(G, 2 * G, dtype=torch.int32, device="cuda")]
    d_out = torch.empty(G, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    h = state._h
    _lib.check(lib.spx_acc_refresh_device(h, d_boards[0].data_ptr(), slots[0].data_ptr(), G, stream))

    def board_index(step):  # 0,1,..,L-1,L-2,..,0,1,..
        k = step % (2 * L - 2)
        return k if k < L else 2 * L - 2 - k

    step_no = [0]

    def step():
        s = step_no[0]
        cur, nxt = slots[s & 1], slots[(s + 1) & 1]
        _lib.check(lib.spx_acc_update_eval_device(h, cur.data_ptr(), nxt.data_ptr(), d_boards[board_index(s + 1)].data_ptr(),
                                                  G, d_out.data_ptr(), stream))
        step_no[0] = s + 1

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    elapsed = group.max_float(time.perf_counter() - t0)
    # parity inside the bench: the incrementally maintained evals equal a full refresh of the final boards
    full = torch.empty(G, dtype=torch.int32, device="cuda")
    state.evaluate_once_device(d_boards[board_index(step_no[0])].data_ptr(), G, full.data_ptr(), stream)
    torch.cuda.synchronize()
    exact = group.sum_int(int(torch.equal(full, d_out))) == world
    if rank == 0:
        psq_d, thr_d = delta_rows_sample(sp, chain[3], chain[4])
        algo = 2048 * psq_d + 1024 * thr_d + 2 * 4096 + 72
        value = world * G * args.steps / elapsed
        print(json.dumps({
            "metric": "nnue_incremental_updates_per_sec", "value": value, "unit": "updates+evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i16 accumulate / i8 MFMA L1 / i32 tail", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[3] shape: per step one ply of incremental accumulator updates "
                                   "(device-derived add/sub deltas) + evaluation for every one of the concurrent games",
                       "games_per_gpu": G, "bit_exact_vs_full_refresh": bool(exact),
                       "mean_delta_rows_per_update": {"psq": psq_d, "threat": thr_d}},
            "roofline": {"kernel": "spx_update_kernel + spx_slot_act_kernel + spx_mlp_kernel", "bound": "hbm",
                         "achieved": algo * value / world / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo * value / world / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_update": algo},
        }), flush=True)
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="positions per GPU per step")
    ap.add_argument("--preset", default="tame", choices=["tame", "wild", "extreme"])
    ap.add_argument("--mode", default="full", choices=["full", "incremental"],
                    help="full = BASELINE configs[1] (headline); incremental = configs[2]/[3] shape: one ply of "
                         "parent->child accumulator updates + evaluation for --batch concurrent games per step")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts/HIP streams the steps are issued round-robin on (2 lets the small sort/MLP kernels of "
                         "one step overlap the texture-bound gather of the next)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="tile this many distinct positions to fill the batch (cache-locality ablation; also keeps host "
                         "generation bounded for the HBM-filling batches of BASELINE config 5)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="issue the steps with the strictly stream-ordered spx_eval_full_device instead of the pipelined "
                         "spx_eval_full_device_async (consecutive batches overlap: sorts / MLP of one beside the "
                         "feature-transformer kernel of the next)")
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
    # Debug knobs for exercising the N > 1 control flow on a single-GPU box (never used by the driver):
    #   SPX_BENCH_SHARE_GPU=1 maps every rank to GPU 0, SPX_BENCH_BACKEND=gloo swaps RCCL for gloo (CPU tensors).
    if os.environ.get("SPX_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")  # "nccl" is RCCL over xGMI on ROCm
    group = Group(backend=backend, device=torch.device("cuda", local_rank) if backend == "nccl" else None)

    if args.mode == "incremental":
        return incremental_bench(args, sp, torch, group, rank, local_rank, world)

    # ---- workload: this rank's shard of seeded random legal positions, resident in HBM ----
    blob = sp.synthetic_net_bytes(args.preset)
    net = sp.Network(blob)
    n_ctx = max(1, args.streams)
    states = [sp.NnueState(net, device=local_rank, max_batch=args.batch) for _ in range(n_ctx)]
    state = states[0]
    n_distinct = min(args.distinct or args.batch, args.batch)
    distinct = sp.random_positions(n_distinct, seed=20260927 + rank, min_ply=8, max_ply=120, dfrc_every=4)
    positions = np.resize(distinct, args.batch) if n_distinct < args.batch else distinct
    d_pos = torch.from_numpy(positions.view(np.uint8).reshape(-1, 32)).cuda()
    pipelined = not args.no_pipeline and n_ctx == 1
    d_outs = [torch.empty(args.batch, dtype=torch.int32, device="cuda") for _ in range(2 if pipelined else n_ctx)]
    d_out = d_outs[0]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_ctx - 1)]
    counter = [0]

    def step():
        k = counter[0] % len(d_outs)
        counter[0] += 1
        if pipelined:  # returns at once; the library chains the batches on its own two streams
            state.evaluate_once_device_async(d_pos.data_ptr(), args.batch, d_outs[k].data_ptr())
        else:
            states[k].evaluate_once_device(d_pos.data_ptr(), args.batch, d_outs[k].data_ptr(), streams[k].cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    for st in states:
        st.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    group.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = [st.profile_end() for st in states]
    sort_ms, ft_ms, mlp_ms, calls = (sum(p[i] for p in prof) for i in range(4))

    elapsed = group.max_float(elapsed)  # slowest rank defines the step time
    checksum = group.sum_int(int(d_out.to(torch.int64).sum().item()))  # checksum of checksums over all shards

    if rank == 0:
        psq_rows, thr_rows = sp.count_rows(distinct)  # host-side count; tiled batches scale the distinct block
        if n_distinct < args.batch:
            psq_rows, thr_rows = (int(v * (args.batch / n_distinct)) for v in (psq_rows, thr_rows))
        algo_bytes = 2048 * psq_rows + 1024 * thr_rows + 36 * args.batch  # per launch
        compact_rows = states[0].compact_psq_rows
        psq_bytes = {11264: 1024, 0: 2048}.get(compact_rows)  # mixed nets: not derivable from the row totals
        requested = None if psq_bytes is None else psq_bytes * psq_rows + 1024 * thr_rows + 36 * args.batch
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
                "workload": (f"BASELINE configs[1]: full-refresh NNUE forward on {args.batch} seeded random legal positions "
                             "per GPU (random playouts 8-120 plies, every 4th game DFRC), bit-exact vs CPU"
                             + (f"; batch tiled from {n_distinct} distinct positions" if n_distinct < args.batch else "")),
                "batch_per_gpu": args.batch,
                "net": f"synthetic CBNF '{net.name}' (Stormphrax 8.0.2 shape: (704x16+64368)->1024)x2->(32x2->64->1)x8",
                "parallelism": f"positions sharded over {world} GPU(s), no collective on the data path; "
                               + ("steps issued through the pipelined spx_eval_full_device_async (two internal streams: "
                                  "sorts / MLP of a batch overlap the next batch's feature-transformer kernel)"
                                  if pipelined else f"steps issued round-robin on {n_ctx} HIP stream(s) per GPU"),
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
                "traffic_note": "fabric+HBM GB/s of the FT kernel from the committed rocprofv3 PMC passes "
                                "(profiles/ft_traffic_pmc.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); null if "
                                "this run's configuration was not profiled",
                "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_position": algo_bytes / args.batch,
                "compact_psq_rows": compact_rows,
                "requested_bytes_per_launch": requested,
                "requested_note": "bytes the kernel's loads ask for: piece-square rows whose weights all fit i8 "
                                  f"({compact_rows} of 11264 in this net) are fetched as 1 KiB u8 copies instead of "
                                  "2 KiB i16 rows (bit-identical sums); `achieved` stays in algorithmic bytes",
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 of the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(sp, positions, blob, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    group.close()


if __name__ == "__main__":
    main()
