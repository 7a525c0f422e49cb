#!/usr/bin/env python3
"""bench.py - batched NNUE position-evals/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" = one full-refresh pass of the hot path (feature extraction + FT accumulation + pairwise activation + int8
MFMA L1 + i32 tail) over ONE batch of synthetic positions already resident in HBM (BASELINE config 2: 65 536 seeded
random legal positions per GPU, synthetic net). N > 1: every rank evaluates its own shard (positions are independent -
no data-path collective), weak scaling; value = all ranks' positions / max-over-ranks time.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 200 --warmup 20

Rank 0 prints ONE JSON line. Extra objects:
  `roofline`      the dominant kernel - the column-sliced pipeline's gather (spx_ftx_gather_kernel; batches below 16 384
                  positions: spx_ft_kernel) - against the roof that BINDS it. The 100 MB of weight rows are resident in L2 /
                  Infinity Cache, so the gather is bound by the L2 -> CU path, not by HBM: `bound` = "l2", `achieved` = bytes
                  the kernel's row loads request from the L2s per launch / its HIP-event duration, `peak` = the 34.5 TB/s
                  aggregate L2 bandwidth. The SURVEY 8(d) algorithmic-bytes figure and the measured HBM/fabric traffic
                  (rocprofv3 PMC pass committed under profiles/) are reported beside it under `hbm`, the VALU issue
                  utilisation from the same PMC passes under `valu`. Every `frac` is <= 1 and recomputable from profiles/.
  `wide_psq_rows` the same timed run on a context with SPX_CTX_WIDE_PSQ_ROWS (the one-kernel path, no 1 KiB u8 copies of
                  piece-square rows): every piece-square row fetched as its 2 KiB i16 row.
  `cpu_baseline`  the compiled reference (`oracle/_ref/sp_ref_probe_tame[_avx512]`) timed on a bounded sample of the same
                  batch on this box's host cores. Its absence is an error (--allow-port-baseline times the scalar C
                  restatement instead).
Before the W warm-up steps the loop is additionally run untimed until the chip's clocks have settled (>= 1 s and three
consecutive 10-step chunks within 2 %), so a short `--steps 20 --warmup 5` run measures the same steady state as a long one.
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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks (before the HIP runtime starts)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
HBM_ACHIEVABLE_GBS = 6300.0  # same guide: "~6.3 TB/s achievable" - the rate the L2-miss path is judged against
L2_PEAK_GBS = 34500.0  # same guide, "L2 (per XCD)": 4 MiB x 8, ~34.5 TB/s aggregate
N_SIMDS = 256 * 4      # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # same guide: peak engine clock
L1_PEAK_GBS = 256 * 64 * CLOCK_HZ / 1e9  # the CU's texture / L1 path: 64 B per clock and CU = 39.3 TB/s
LDS_PEAK_GBS = 256 * 128 * CLOCK_HZ / 1e9  # LDS: 128 B per clock and CU = 78.6 TB/s
# tools/probes/tcp_mask_probe.hip (profiles/r05_probe_texture_path_cost_of_masked_and_narrow_loads.txt): cycles one wave instruction
# of 16 bytes per lane holds the texture path / the LDS pipe of its CU, whatever its exec mask or width
L1_CYCLES_PER_WAVE_LOAD, LDS_CYCLES_PER_WAVE_READ, MFMA_CYCLES = 17.5, 8.8, 16.0
# committed rocprofv3 PMC passes of this command, newest round first (tools/gpu_final.sh puts this round's in place before the
# bench lines are taken; the kernels these counters describe did not change in round 3)
PMC_FILES = {"full": ["r06_pmc_full_refresh.json", "r05_pmc_full_refresh.json", "r04_pmc_full_refresh.json", "r03_pmc_full_refresh.json", "r02_pmc_full_refresh.json"],
             "incremental": ["r06_pmc_incremental.json", "r05_pmc_incremental.json", "r04_pmc_incremental.json", "r03_pmc_incremental.json", "r02_pmc_incremental.json"]}


def settle(step, sync, min_seconds=1.0, max_seconds=8.0, chunk=10):
    """Untimed clock / cache warm-up: run `chunk`-step groups until >= min_seconds have passed AND the last three group
    times agree within 2 % (DVFS has settled), or max_seconds. Returns the number of steps issued."""
    times, total, steps = [], 0.0, 0
    while total < max_seconds:
        t0 = time.perf_counter()
        for _ in range(chunk):
            step()
        sync()
        dt = time.perf_counter() - t0
        if dt > 2.0 * min_seconds:  # steps that take seconds themselves (HBM-filling batches): one group is warm-up enough
            return steps + chunk
        if dt > 0.25 and chunk > 1:
            chunk = 1
        times.append(dt)
        total += dt
        steps += chunk
        if total >= min_seconds and len(times) >= 3 and max(times[-3:]) <= 1.02 * min(times[-3:]):
            break
    return steps


def load_pmc(mode, **match):
    """Counters of the committed rocprofv3 PMC passes of this command (profiles/r0N_pmc_*.json, written by
    tools/pmc_to_json.py from the passes of tools/gpu_profile.sh). None when this run's configuration was not profiled."""
    for name in PMC_FILES[mode]:
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if all(rec.get("config", {}).get(k) == v for k, v in match.items()):
            rec["file"] = "profiles/" + name
            return rec
    return None


def kernel_pmc(rec, name):
    if not rec:
        return None
    for k, v in rec.get("kernels", {}).items():
        if name in k:
            return v
    return None


def valu_block(kp, cycles_per_instr):
    """VALU issue utilisation of one kernel from its PMC pass: wave-level VALU instructions x SIMD cycles per instruction
    / (SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8."""
    if not kp or "SQ_INSTS_VALU" not in kp.get("counters", {}) or "GRBM_GUI_ACTIVE" not in kp["counters"]:
        return None
    c = kp["counters"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    return {"insts_per_launch": c["SQ_INSTS_VALU"], "kernel_cycles": cycles, "cycles_per_wave_instr": cycles_per_instr,
            "util": c["SQ_INSTS_VALU"] * cycles_per_instr / (N_SIMDS * cycles),
            "source": "profiles/ (rocprofv3 --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE passes of this command)"}


def oracle_sample_check(sp, blob, positions, got):
    """Parity inside the bench, outside the timed region: the scalar C restatement (oracle/, test infrastructure) on a
    sample of the batch vs the GPU's scores for the same positions."""
    so = os.path.join(ROOT, "oracle", "libspx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
    oracle = ctypes.CDLL(so)
    oracle.spxo_init.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    oracle.spxo_eval_mailboxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert oracle.spxo_init(blob.ctypes.data, blob.size) == 0
    mail, stm = sp.positions_to_mailboxes(positions)
    want = np.empty(len(positions), dtype=np.int32)
    assert oracle.spxo_eval_mailboxes(mail.ctypes.data, stm.ctypes.data, len(positions), want.ctypes.data) == 0
    return bool(np.array_equal(want, np.asarray(got, dtype=np.int32)))


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


def cpu_baseline(sp, positions, blob, seconds, allow_port=False):
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
    if not os.path.exists(probe) and not allow_port:
        raise SystemExit(f"bench.py: {probe} is missing - the cpu_baseline leg times the COMPILED REFERENCE, which is built "
                         "in the authoring container (`make -C oracle ref ref512`, needs /root/reference) and travels "
                         "untracked in oracle/_ref/. Re-run with --allow-port-baseline to time the scalar C restatement "
                         "(kind 'port', 1 thread) instead, or with --no-cpu-baseline.")
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
        except Exception as exc:
            if not allow_port:
                raise SystemExit(f"bench.py: the reference probe {probe} failed ({exc}); --allow-port-baseline would time "
                                 "the scalar C restatement instead")
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

    pipelined = not args.no_pipeline
    d_outs = [d_out, torch.empty(G, dtype=torch.int32, device="cuda")]

    def step():
        s = step_no[0]
        cur, nxt = slots[s & 1], slots[(s + 1) & 1]
        boards = d_boards[board_index(s + 1)].data_ptr()
        if pipelined:  # update kernels chained in call order, sort + MLP of a ply beside the next ply's update
            _lib.check(lib.spx_acc_update_eval_device_async(h, cur.data_ptr(), nxt.data_ptr(), boards, G,
                                                            d_outs[s & 1].data_ptr(), None))
        else:
            _lib.check(lib.spx_acc_update_eval_device(h, cur.data_ptr(), nxt.data_ptr(), boards, G,
                                                      d_outs[s & 1].data_ptr(), stream))
        step_no[0] = s + 1

    def sync():
        state.synchronize()
        torch.cuda.synchronize()

    settle_steps = 0 if args.no_settle else settle(step, sync)
    for _ in range(args.warmup):
        step()
    sync()
    group.barrier()
    sync()
    state.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    group.barrier()
    sync()
    elapsed = group.max_float(time.perf_counter() - t0)
    _, update_ms, mlp_ms, calls = state.profile_end()
    # parity inside the bench: the incrementally maintained evals equal a full refresh of the final boards
    full = torch.empty(G, dtype=torch.int32, device="cuda")
    state.evaluate_once_device(d_boards[board_index(step_no[0])].data_ptr(), G, full.data_ptr(), stream)
    torch.cuda.synchronize()
    exact = group.sum_int(int(torch.equal(full, d_outs[(step_no[0] - 1) & 1]))) == world
    if rank == 0:
        psq_d, thr_d = delta_rows_sample(sp, chain[3], chain[4])
        compulsory = 2 * 4096 + 72  # parent accumulators read + child accumulators written + the two records
        update_s = update_ms / max(calls, 1) / 1e3
        value = world * G * args.steps / elapsed
        pmc = load_pmc("incremental", batch=G, preset=args.preset)
        kp = kernel_pmc(pmc, "spx_update_kernel<")
        roofline = {
            "kernel": "spx_update_kernel (+ its rebuild pass)", "bound": "hbm",
            "achieved": compulsory * G / update_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": compulsory * G / update_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "update_kernel_ms": update_s * 1e3, "sort_mlp_ms": mlp_ms / max(calls, 1),
            "compulsory_bytes_per_update": compulsory,
            "note": "achieved = the bytes an update MUST move through HBM (parent accumulators 4 KiB in, child "
                    "accumulators 4 KiB out, records) x records / the HIP-event duration of the update kernel and its "
                    "rebuild pass; the delta rows (mean_delta_rows_per_update x 1-2 KiB) come from L2 / Infinity Cache "
                    "and are reported as gather_bytes_per_update, not counted against the HBM roof. traffic = measured "
                    "fabric+HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE of the committed PMC pass)",
            "gather_bytes_per_update": 2048 * psq_d + 1024 * thr_d, "valu": None,
        }
        if kp and "FETCH_SIZE" in kp["counters"] and "WRITE_SIZE" in kp["counters"]:
            roofline["traffic"] = (2 * kp["counters"]["FETCH_SIZE"] + kp["counters"]["WRITE_SIZE"]) * 1024
            # what actually crosses the fabric (arena + the delta rows that miss the L2s; Infinity-Cache hits included, so an
            # upper bound on HBM bytes) against the same HBM peak: the figure that says how close the kernel is to a memory roof
            roofline["traffic_gbs"] = roofline["traffic"] / update_s / 1e9
            roofline["traffic_over_hbm_peak"] = roofline["traffic_gbs"] / HBM_PEAK_GBS
            roofline["traffic_over_achievable"] = roofline["traffic_gbs"] / HBM_ACHIEVABLE_GBS  # the binding path
        if pmc:
            roofline["valu"] = valu_block(kp, pmc.get("valu_cycles_per_wave_instr", 4))
        print(json.dumps({
            "metric": "nnue_incremental_updates_per_sec", "value": value, "unit": "updates+evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i16 accumulate / i8 MFMA L1 / i32 tail", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[3] shape: per step one ply of incremental accumulator updates "
                                   "(device-derived add/sub deltas) + evaluation for every one of the concurrent games",
                       "games_per_gpu": G, "bit_exact_vs_full_refresh": bool(exact), "settle_steps": settle_steps,
                       "issue": ("pipelined spx_acc_update_eval_device_async: update kernels in call order, sort + MLP of a "
                                 "ply beside the next ply's update kernel" if pipelined else "stream-ordered calls"),
                       "mean_delta_rows_per_update": {"psq": psq_d, "threat": thr_d}},
            "roofline": roofline,
        }), flush=True)
    group.close()


def timed_full_run(args, torch, group, state, d_pos, pipelined, n_outs=2, settle_seconds=1.0):
    """Settle, W warm-up steps, then exactly K timed steps of the full-refresh path on `state`.
    -> (elapsed max over ranks, per-kernel ms (sort, ft, mlp, calls), settle steps, the output tensor of the last step)."""
    # (a batch above the scratch capacity is pipelined chunk by chunk inside ONE call: a second output buffer buys nothing)
    many = n_outs if pipelined and args.batch <= state.scratch_batch else 1
    d_outs = [torch.empty(args.batch, dtype=torch.int32, device="cuda") for _ in range(many)]
    stream = torch.cuda.current_stream().cuda_stream
    counter = [0]

    def step():
        k = counter[0] % len(d_outs)
        counter[0] += 1
        if pipelined:  # returns at once; the library chains the batches on its own two streams
            state.evaluate_once_device_async(d_pos.data_ptr(), args.batch, d_outs[k].data_ptr())
        else:
            state.evaluate_once_device(d_pos.data_ptr(), args.batch, d_outs[k].data_ptr(), stream)

    def sync():
        state.synchronize()
        torch.cuda.synchronize()

    settle_steps = 0 if args.no_settle else settle(step, sync, min_seconds=settle_seconds, chunk=10 if args.batch <= (1 << 22) else 1)
    for _ in range(args.warmup):
        step()
    sync()
    group.barrier()
    sync()
    # calls above the scratch capacity run as several launch sequences; pipelined calls above one pass of the column-sliced
    # pipeline (65 536 positions) too, one per pass
    per_call = state.scratch_batch
    if pipelined and state.takes_sliced_pipeline(args.batch, pipelined=True):
        per_call = min(per_call, 65536)
    chunks = -(-args.batch // per_call)
    state.profile_begin(min(args.steps * chunks, 1 << 16))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    group.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    sort_ms, ft_ms, mlp_ms, calls = state.profile_end()
    if os.environ.get("SPX_BENCH_DIAG"):  # the last gather's own clock (workgroup start / end stamps) beside the events' interval
        from stormphrax_amd import _lib
        spans = []
        for slot in ((0, 1, 2) if pipelined else (-1,)):
            t = np.zeros(512, dtype=np.uint64)
            if _lib.load().spx_debug_ftx_block_times(state._h, slot, t.ctypes.data) == 0:
                start, end = t[0::2].astype(np.int64), t[1::2].astype(np.int64)
                spans.append(round(float(end.max() - start.min()) / 100.0, 1))
        print(f"[diag] gather span by its own clock (us) {spans}; events: ft {ft_ms / max(calls, 1) * 1e3:.1f} us, mlp {mlp_ms / max(calls, 1) * 1e3:.1f} us; "
              f"step {elapsed / args.steps * 1e6:.1f} us", file=sys.stderr, flush=True)
    timed_full_run.prepare_ms = state.profile_prepare_ms() / max(calls / chunks, 1)  # per step; see spx_profile_last_prepare_ms
    prof = (sort_ms, ft_ms, mlp_ms, calls / chunks)  # per-kernel times per STEP (= per batch), as the byte counts are
    # per-rank view for the report (rank order): each rank's FT-kernel time is its own, free of the barrier wait
    timed_full_run.per_rank_ft_ms = group.all_floats(ft_ms / max(calls / chunks, 1))
    return group.max_float(elapsed), prof, settle_steps, d_outs[(counter[0] - 1) % len(d_outs)]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (one process per GPU under
    torch.distributed.run on 127.0.0.1, a free port) with the same arguments, pass their output through and return
    their exit status - the same command shape works for N = 1 and N > 1."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE in the environment: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


def realistic_leg(args, sp, torch, group, d_pos, positions, pipelined):
    """Third headline: the same timed run on the `realistic` preset (heavy-tailed weights: ~42 % of the piece-square rows fit
    i8, ~38 % have <= 32 weights outside it, ~20 % are wide) - what a trained net should expect rather than the all-compact
    best case of the uniform presets. Scores checked against the CPU oracle on a sample."""
    blob = sp.synthetic_net_bytes("realistic")
    net = sp.Network(blob)
    st = sp.NnueState(net, device=torch.cuda.current_device(), max_batch=args.batch)
    try:
        elapsed, (_, ft_ms, _, calls), _, last = timed_full_run(args, torch, group, st, d_pos, pipelined, settle_seconds=0.4)
        n_sample = min(1024, len(positions))
        exact = oracle_sample_check(sp, blob, positions[:n_sample], last[:n_sample].cpu().numpy())
        fit, near, wide = net.psq_row_classes()
        if st.takes_sliced_pipeline(min(args.batch, st.scratch_batch), pipelined=pipelined):
            # what the gather walked (pack kernel's counts): every piece-square row outside i8 - near-compact ones too - brings a
            # high-byte plane through the texture path
            walk = st.ftx_walk(0 if pipelined else -1)
            rows = {"through_the_texture_path_1KiB": walk["global_rows"], "from_lds_1KiB": walk["lds_rows"],
                    "global_steps_per_slice": walk["global_steps"], "lds_steps_per_slice": walk["lds_steps"]}
        else:
            wide_rows, compact_rows, thr_rows = st.count_rows(positions)
            rows = {"psq_wide_2KiB": wide_rows, "psq_1KiB_compact_or_near": compact_rows, "threat_1KiB": thr_rows}
        return {"value": args.batch * args.steps / elapsed, "unit": "evals/s", "ms_per_step": elapsed / args.steps * 1e3,
                "ft_kernel_ms": ft_ms / max(calls, 1), "bit_exact_sample": bool(exact),
                "psq_row_classes": {"fit_i8": fit, "near_compact_le_32_outliers": near, "wide": wide},
                "rows_per_launch": rows,
                "net": f"synthetic CBNF '{net.name}': two-sided geometric (Laplace-like) weights with heavy tails, the shape of a "
                       "trained QA = 255 net (spx_synth.cpp preset 3); reference goldens in tests/golden/evals.jsonl"}
    finally:
        st.close()


def paths_leg(args, sp, torch, group, net, blob, d_pos, positions):
    """secondary.full_refresh_paths: the headline batch through BOTH full-refresh implementations - the column-sliced pipeline
    (stormphrax_amd/csrc/spx_ftx.hip; the default from 16 384 positions up) and the one-kernel path (spx_ft_kernel:
    SPX_CTX_ONE_KERNEL_FT) - stream-ordered and pipelined, with the main kernel's own time and what runs before it. Scores checked
    against the CPU oracle on a sample; checksums must agree between the paths. (Round 4 also timed the pipeline with its lists NOT
    rebuilt - the bound a free preparation would give, 1.95e8 - through a switch inside the library; the switch is gone)"""
    import copy

    out = {}
    for path, mode in (("sliced_pipeline", "stream_ordered"), ("sliced_pipeline", "pipelined"),
                       ("one_kernel", "stream_ordered"), ("one_kernel", "pipelined")):
        st = sp.NnueState(net, device=torch.cuda.current_device(), max_batch=args.batch, sliced_ft=path == "sliced_pipeline")
        try:
            a = copy.copy(args)
            a.steps, a.warmup = min(args.steps, 100), min(args.warmup, 10)
            elapsed, (sort_ms, ft_ms, mlp_ms, calls), _, last = timed_full_run(a, torch, group, st, d_pos, mode == "pipelined",
                                                                              settle_seconds=0.3)
            calls = max(calls, 1)
            rec = {"value": args.batch * a.steps / elapsed, "unit": "evals/s", "ms_per_step": elapsed / a.steps * 1e3,
                   "main_kernel": "spx_ftx_gather_kernel" if path == "sliced_pipeline" else "spx_ft_kernel",
                   "main_kernel_ms": ft_ms / calls, "sort_ms": sort_ms / calls, "mlp_ms": mlp_ms / calls}
            if mode == "stream_ordered":
                rec["before_main_kernel_ms"] = timed_full_run.prepare_ms  # the pipeline's extraction, counting sort, plan
                n_sample = min(2048, len(positions))
                rec["bit_exact_sample"] = bool(oracle_sample_check(sp, blob, positions[:n_sample], last[:n_sample].cpu().numpy()))
                rec["checksum"] = int(last.sum(dtype=torch.int64).item())
            out.setdefault(path, {})[mode] = rec
        finally:
            st.close()
    out["checksums_agree"] = out["sliced_pipeline"]["stream_ordered"]["checksum"] == out["one_kernel"]["stream_ordered"]["checksum"]
    out["note"] = ("sliced pipeline: XCD x reads the 128-byte slice x of every row (L2 hit rate 76 -> 83-89 %, fabric bytes / 3), "
                   "the king bucket's piece-square slab lives in LDS, four gathered i8 rows are widened and added by ONE "
                   "v_mfma_i32_16x16x64_i8: the gather takes 0.7 x spx_ft_kernel's time. The row lists must cross XCDs, i.e. be "
                   "produced by a pass of their own (one extraction per position, a counting sort by (king bucket, list length), a "
                   "plan); kernels sharing the CUs slow each other down by what they would take alone, so the step costs the sum "
                   "(DESIGN.md 4.9)")
    return out


def incremental_leg(sp, torch, net, device, games=65536, chain=6, seconds=0.6):
    """secondary.incremental (BASELINE configs[2] shape): `games` concurrent games, one fused update + eval batch per ply,
    pipelined plies. Boards are generated on the device: spx_random_positions_gpu is deterministic per seed, so the same
    call with one more ply yields each position's successor; four material phases (10 / 30 / 60 / 100 plies)."""
    from stormphrax_amd import _lib

    lib = _lib.load()
    st = sp.NnueState(net, device=device, max_batch=games)
    try:
        boards = []
        quarter = games // 4
        for k in range(chain):
            t = torch.empty((games, 32), dtype=torch.uint8, device="cuda")
            for q, base in enumerate((10, 30, 60, 100)):
                cnt = quarter if q < 3 else games - 3 * quarter
                st.random_positions_device(t[q * quarter:].data_ptr(), cnt, seed=4242 + q, min_ply=base + k, max_ply=base + k,
                                           dfrc_every=4)
            boards.append(t)
        sample = [b[:: max(1, games // 256)].cpu().numpy().reshape(-1).view(sp.PACKED_DTYPE) for b in boards[:2]]
        changed = [int(np.count_nonzero(sp.positions_to_mailboxes(sample[0][i:i + 1])[0] != sp.positions_to_mailboxes(sample[1][i:i + 1])[0]))
                   for i in range(len(sample[0]))]
        st.reserve_slots(2 * games)
        slots = [torch.arange(games, dtype=torch.int32, device="cuda"), torch.arange(games, 2 * games, dtype=torch.int32, device="cuda")]
        outs = [torch.empty(games, dtype=torch.int32, device="cuda") for _ in range(2)]
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.spx_acc_refresh_device(st._h, boards[0].data_ptr(), slots[0].data_ptr(), games, stream))
        torch.cuda.synchronize()
        L, no = chain, [0]

        def board_index(step):  # 0, 1, .., L-1, L-2, .., 0, 1, ..: an unmake is a one-move delta too
            k = step % (2 * L - 2)
            return k if k < L else 2 * L - 2 - k

        def step():
            s_ = no[0]
            _lib.check(lib.spx_acc_update_eval_device_async(st._h, slots[s_ & 1].data_ptr(), slots[(s_ + 1) & 1].data_ptr(),
                                                            boards[board_index(s_ + 1)].data_ptr(), games,
                                                            outs[s_ & 1].data_ptr(), None))
            no[0] = s_ + 1

        def sync():
            st.synchronize()
            torch.cuda.synchronize()

        settle(step, sync, min_seconds=0.3, max_seconds=2.0)
        steps = 0
        st.profile_begin(4096)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds and steps < 4000:
            for _ in range(20):
                step()
            steps += 20
            sync()
        elapsed = time.perf_counter() - t0
        _, update_ms, mlp_ms, calls = st.profile_end()
        full = torch.empty(games, dtype=torch.int32, device="cuda")
        st.evaluate_once_device(boards[board_index(no[0])].data_ptr(), games, full.data_ptr(), stream)
        torch.cuda.synchronize()
        exact = bool(torch.equal(full, outs[(no[0] - 1) & 1]))
        compulsory = 2 * 4096 + 72
        update_s = update_ms / max(calls, 1) / 1e3
        return {"value": games * steps / elapsed, "unit": "updates+evals/s", "games": games, "steps": steps,
                "ms_per_step": elapsed / steps * 1e3, "update_kernel_ms": update_s * 1e3, "sort_mlp_ms": mlp_ms / max(calls, 1),
                "bit_exact_vs_full_refresh": exact, "squares_changed_per_move_sample_max": max(changed),
                "roofline": {"kernel": "spx_update_kernel (+ its rebuild pass)", "bound": "hbm",
                             "achieved": compulsory * games / update_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": compulsory * games / update_s / 1e9 / HBM_PEAK_GBS,
                             "note": "compulsory bytes (parent accumulators in, child accumulators out, records) x records / "
                                     "HIP-event time of the update kernel and its rebuild pass; PMC traffic: profiles/"}}
    finally:
        st.close()


def siblings_leg(sp, torch, net, device, parents=2048, seconds=0.5):
    """secondary.incremental_siblings (the other SHAPE of the incremental path, reported beside
    secondary.incremental): `parents` positions with resident accumulators and ALL their legal children (~33 each) updated
    and evaluated as eval-only children - one parent accumulator read per ~33 records (they are neighbours in the batch), no child
    accumulator written. This is what the self-play drivers and a search's node expansion issue; secondary.incremental is the
    other extreme (every record its own parent, every child stored: 65 536 independent games one ply on)."""
    from stormphrax_amd import _lib

    lib = _lib.load()
    pos = sp.random_positions(parents, seed=777, min_ply=8, max_ply=120, dfrc_every=4)
    st = sp.NnueState(net, device=device, max_batch=parents * 96)
    try:
        st.reserve_slots(parents)
        slot_ids = np.arange(parents, dtype=np.uint32)
        mg = st.movegen(pos, parent_values=slot_ids, capacity=parents * 96)
        n = int(mg["count"].sum())
        d_pos = torch.from_numpy(pos.view(np.uint8).reshape(parents, 32).copy()).cuda()
        d_slots = torch.from_numpy(slot_ids.astype(np.int32)).cuda()
        d_children = torch.from_numpy(mg["children"][:n].view(np.uint8).reshape(n, 32).copy()).cuda()
        d_parents = torch.from_numpy(mg["parents"][:n].astype(np.int32)).cuda()
        outs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.spx_acc_refresh_device(st._h, d_pos.data_ptr(), d_slots.data_ptr(), parents, stream))
        torch.cuda.synchronize()
        no = [0]

        def step():
            _lib.check(lib.spx_acc_update_eval_device_async(st._h, d_parents.data_ptr(), None, d_children.data_ptr(), n,
                                                            outs[no[0] & 1].data_ptr(), None))
            no[0] += 1

        def sync():
            st.synchronize()
            torch.cuda.synchronize()

        settle(step, sync, min_seconds=0.2, max_seconds=1.5)
        steps = 0
        st.profile_begin(4096)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds and steps < 4000:
            for _ in range(10):
                step()
            steps += 10
            sync()
        elapsed = time.perf_counter() - t0
        _, update_ms, mlp_ms, calls = st.profile_end()
        full = torch.empty(n, dtype=torch.int32, device="cuda")
        st.evaluate_once_device(d_children.data_ptr(), n, full.data_ptr(), stream)
        torch.cuda.synchronize()
        return {"value": n * steps / elapsed, "unit": "updates+evals/s", "parents": parents, "children": n,
                "children_per_parent": n / parents, "ms_per_step": elapsed / steps * 1e3,
                "update_kernel_ms": update_ms / max(calls, 1), "sort_mlp_ms": mlp_ms / max(calls, 1),
                "bit_exact_vs_full_refresh": bool(torch.equal(full, outs[0]) and torch.equal(full, outs[1]))}
    finally:
        st.close()


def config3_leg(sp, net, device, name="trace_startpos_tame_64k.txt.gz",
                what="recorded from the compiled reference: depth-12 make/unmake walk from the start position through "
                     "NnueState::push / pop / evaluate"):
    """secondary.config3_replay: a recorded 65 536-EVAL reference trace (tests/golden, BASELINE configs[2]) through ONE
    native spx_acc_replay_tree call; every EVAL must equal what the reference's NnueState::evaluate recorded."""
    from stormphrax_amd.trace import Trace, replay_native

    path = os.path.join(ROOT, "tests", "golden", name)
    trace = Trace(path)
    st = sp.NnueState(net, device=device, max_batch=65536)
    try:
        pos = trace.positions()
        got, want, ms = replay_native(st, trace, pos)
        got2, _, ms2 = replay_native(st, trace, pos)
        return {"device_ms": min(ms, ms2), "updates": trace.n_nodes - 1, "evals": len(want),
                "updates_plus_evals_per_sec": (trace.n_nodes - 1 + len(want)) / (min(ms, ms2) / 1e3),
                "every_eval_equals_the_reference": bool(np.array_equal(got, want) and np.array_equal(got2, want)),
                "tree_levels": int(max(trace.depth)), "trace": f"tests/golden/{name} ({what})"}
    finally:
        st.close()


def forest_leg(sp, net, device, name="forest_search_256x1024_tame.npz"):
    """secondary.config3_alpha_beta_replay_x256: 256 alpha-beta search trees of the reference (its own
    search recorded from 256 different roots, tests/golden/make_golden.py `forest`) in flight AT ONCE through ONE
    spx_acc_replay_tree call - the shape thousands of concurrent searches give the evaluator, where the single tree of
    config3_alpha_beta_replay is a latency figure. Every EVAL must equal what the reference's NnueState::evaluate recorded."""
    from stormphrax_amd.trace import Forest, replay_forest

    forest = Forest(os.path.join(ROOT, "tests", "golden", name))
    st = sp.NnueState(net, device=device, max_batch=1 << 17)
    try:
        pos = forest.positions()
        runs = [replay_forest(st, forest, pos) for _ in range(3)]
        ms = min(r[2] for r in runs)
        ok = all(bool(np.array_equal(r[0], forest.eval_value)) for r in runs)
        work = forest.n_nodes - 1 + len(forest.eval_node)
        return {"value": work / (ms / 1e3), "unit": "updates+evals/s", "device_ms": ms, "trees": forest.n_trees,
                "updates": forest.n_nodes - 1, "evals": len(forest.eval_node), "tree_levels": int(forest.depth.max()),
                "every_eval_equals_the_reference": ok,
                "trace": f"tests/golden/{name} (the reference's own alpha-beta search, depth <= 12, from {forest.n_trees} roots: "
                         "random playouts of 6-60 plies, every second one double Chess960)"}
    finally:
        st.close()


def selfplay_leg(sp, net, device, seats=4096, games=32768, seed=1):
    """secondary.config4_selfplay (BASELINE configs[3] shape on one GPU): `seats` concurrent games living on the device, every
    legal move of every game evaluated per ply (eval-only children), the reference's datagen rules in the step kernel; games
    are discarded (out_path None). The verification of such files - every game replayed through the restated rules, samples
    against the oracle - is tests/ and tools/gpu_selfplay_soak.sh, not this leg."""
    st = sp.NnueState(net, device=device, max_batch=seats * 64)
    try:
        stats = st.selfplay(n_games=seats, target_games=games, out_path=None, max_plies=300, dfrc=True, temperature_cp=30, seed=seed)
        return {"value": stats["evals"] / stats["seconds"], "unit": "leaf evals/s", "concurrent_games": seats, "evals": stats["evals"],
                "games": stats["games"], "positions": stats["positions"], "seconds": stats["seconds"],
                "gpu_call_fraction": stats["gpu_seconds"] / stats["seconds"], "outcomes_white_loss_draw_win": stats["outcomes"],
                "policy": "depth-1: score(move) = -NNUE(child), uniform among the moves within 30 cp of the best; openings, "
                          "verification filter, adjudication, Position::isDrawn, viriformat records as src/datagen/datagen.cpp"}
    finally:
        st.close()


def selfplay_search_leg(sp, net, device, seats=4096, games=4096, nodes=64, seed=1):
    """secondary.config4_selfplay_search: the same device-resident games with a LIVE
    fixed-node search in place of the depth-1 policy - every seat runs its own iterative-deepening alpha-beta, one node
    expanded per seat and round (spx_search_step_kernel), the node's children evaluated together with every other seat's.
    `nodes` = expansions after which a search stops at the end of an iteration. tests/test_gpu_incremental.py replays such
    games through a recursive restatement of the search (tests/_search_rules.py); this leg only measures."""
    st = sp.NnueState(net, device=device, max_batch=seats * 64)
    try:
        stats = st.selfplay(n_games=seats, target_games=games, out_path=None, max_plies=300, dfrc=True, temperature_cp=0, seed=seed,
                            search_nodes=nodes)
        return {"value": stats["evals"] / stats["seconds"], "unit": "leaf evals/s", "concurrent_games": seats, "evals": stats["evals"],
                "node_budget": nodes, "nodes_expanded": stats["steps"], "games": stats["games"], "positions": stats["positions"],
                "nodes_per_move": stats["steps"] / max(1, stats["positions"]), "leaves_per_node": stats["evals"] / max(1, stats["steps"]),
                "seconds": stats["seconds"], "gpu_call_fraction": stats["gpu_seconds"] / stats["seconds"],
                "outcomes_white_loss_draw_win": stats["outcomes"],
                "policy": "iterative-deepening alpha-beta over explicit per-seat stacks, leaves = NNUE(child) through the fused "
                          "eval-only update, children ordered by value, no transposition table; rules in csrc/spx_kernels.h "
                          "(SearchStepParams); openings, verification filter, adjudication, Position::isDrawn, viriformat "
                          "records as src/datagen/datagen.cpp"}
    finally:
        st.close()


def secondary_legs_multi(sp, torch, group, net, device, rank, world, preset):
    """N > 1: BASELINE configs[3] as it is worded - 4 096 concurrent self-play games SHARDED over the GPUs (rank r plays
    its share of the seats and of the target with its own seed: game_id mod N), no collective in the game loop -
    plus the same with 4 096 games PER GPU (weak scaling) and the incremental leg on every rank. Every rank runs every leg
    behind a barrier; rank 0 reports SUM of the work over MAX of the time, and every rank's own figure. Outside the headline's
    timed region. A leg that fails on any rank reports the error instead of taking the line down; the collectives below are
    called by every rank whatever happened."""
    out = {}

    def run(name, fn, reduce):
        group.barrier()
        t0 = time.perf_counter()
        err, mine = "", None
        try:
            mine = fn()
        except Exception as exc:  # noqa: BLE001 - reported, not swallowed
            err = f"rank {rank}: {type(exc).__name__}: {exc}"
        failed = group.sum_int(1 if err else 0)
        seconds = group.max_float(time.perf_counter() - t0)
        if failed:
            out[name] = {"error": err or f"{failed} other rank(s) failed", "ranks_failed": failed}
        else:
            out[name] = reduce(mine)
        out[name]["leg_seconds"] = seconds
        out[name]["n_gpus"] = world

    def reduce_selfplay(m):
        evals = group.sum_int(m["evals"])
        secs = group.all_floats(m["seconds"])
        rates = group.all_floats(m["value"])
        frac = group.all_floats(m["gpu_call_fraction"])
        games = group.sum_int(m["games"])
        positions = group.sum_int(m["positions"])
        return {"value": evals / max(secs), "unit": "leaf evals/s (all GPUs: evals of every rank / the slowest rank's seconds)",
                "concurrent_games_per_gpu": m["concurrent_games"], "concurrent_games": m["concurrent_games"] * world,
                "games": games, "positions": positions, "seconds_per_rank": secs, "leaf_evals_per_sec_per_rank": rates,
                "gpu_call_fraction_per_rank": frac, "sharding": "rank r plays seats and target share r with seed 1 + r; no collective in the game loop",
                "policy": m["policy"]}

    def reduce_incremental(m):
        rates = group.all_floats(m["value"])
        exact = group.sum_int(int(bool(m["bit_exact_vs_full_refresh"]))) == world
        upd = group.all_floats(m["update_kernel_ms"])
        return {"value": float(sum(rates)), "unit": "updates+evals/s (sum of the ranks' rates over the same window)",
                "games_per_gpu": m["games"], "updates_plus_evals_per_sec_per_rank": rates, "update_kernel_ms_per_rank": upd,
                "bit_exact_vs_full_refresh": bool(exact)}

    seats = max(256, 4096 // world)
    run("config4_selfplay", lambda: selfplay_leg(sp, net, device, seats=seats, games=max(2048, 32768 // world), seed=1 + rank),
        reduce_selfplay)
    run("config4_selfplay_4096_games_per_gpu", lambda: selfplay_leg(sp, net, device, seats=4096, games=32768, seed=1 + rank),
        reduce_selfplay)
    if preset == "tame":
        run("incremental", lambda: incremental_leg(sp, torch, net, device), reduce_incremental)
    return out


def secondary_legs(args, sp, torch, group, state, net, d_pos, positions, pipelined, device):
    """Outside the headline's timed region, single-GPU runs only: puts the other configurations on the driver's clock
    (VERDICT r2 item 6). A leg that fails reports its error instead of taking the headline down."""
    out = {}

    def run(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 - reported, not swallowed
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        out[name]["leg_seconds"] = time.perf_counter() - t0

    run("realistic_rows", lambda: realistic_leg(args, sp, torch, group, d_pos, positions, pipelined))
    if args.batch <= state.scratch_batch and args.batch >= 16384 and not args.net:
        run("full_refresh_paths", lambda: paths_leg(args, sp, torch, group, net, sp.synthetic_net_bytes(args.preset), d_pos, positions))
    if args.preset == "tame":  # (the trace was recorded on the tame net)
        run("incremental", lambda: incremental_leg(sp, torch, net, device))
        run("incremental_siblings", lambda: siblings_leg(sp, torch, net, device))
        run("config3_replay", lambda: config3_leg(sp, net, device))
        run("config3_alpha_beta_replay", lambda: config3_leg(
            sp, net, device, "trace_search_startpos_tame_64k.txt.gz",
            "the reference's own alpha-beta search, depth <= 12 from the start position, recorded through link-time interposition"))
        run("config3_alpha_beta_replay_x256", lambda: forest_leg(sp, net, device))
    run("config4_selfplay", lambda: selfplay_leg(sp, net, device))
    run("config4_selfplay_search", lambda: selfplay_search_leg(sp, net, device))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="positions per GPU per step")
    ap.add_argument("--preset", default="tame", choices=["tame", "wild", "extreme", "realistic"])
    ap.add_argument("--net", default=None,
                    help="CBNF net file (plain or zstd-compressed, e.g. Stormphrax's net093_255_128_q6.nnue) instead of the "
                         "synthetic preset; the compiled-reference CPU leg embeds the synthetic net, so it is skipped")
    ap.add_argument("--mode", default="full", choices=["full", "incremental"],
                    help="full = BASELINE configs[1] (headline); incremental = configs[2]/[3] shape: one ply of "
                         "parent->child accumulator updates + evaluation for --batch concurrent games per step")
    ap.add_argument("--distinct", type=int, default=0,
                    help="tile this many distinct positions to fill the batch (cache-locality ablation; also keeps host "
                         "generation bounded for the HBM-filling batches of BASELINE config 5)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="issue the steps with the strictly stream-ordered spx_eval_full_device instead of the pipelined "
                         "spx_eval_full_device_async (consecutive batches overlap: sorts / MLP of one beside the "
                         "feature-transformer kernel of the next)")
    ap.add_argument("--device-positions", action="store_true",
                    help="generate the batch with spx_random_positions_gpu (random playouts on the device: move generation + "
                         "uniform move choice kernels) instead of the host chess core - for nodes whose ranks share few CPUs")
    ap.add_argument("--host-positions", action="store_true",
                    help="N > 1 generates positions on the device by default; this keeps the host chess core (8 s per rank)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1 all-gathers the last step's scores by default; skip it")
    ap.add_argument("--no-settle", action="store_true", help="skip the time-based clock warm-up before the W warm-up steps")
    ap.add_argument("--no-wide", action="store_true", help="skip the second timed run with SPX_CTX_WIDE_PSQ_ROWS")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: all_gather the per-rank scores of the last step (RCCL) and check the gathered array "
                         "against each rank's own shard checksum (SURVEY 8e's optional result gather; outside the timed region)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (incremental ply, config-3 replays, self-play, realistic-weights net, gather ceiling) that "
                         "the single-GPU run appends to the JSON line after the headline's timed region")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-port-baseline", action="store_true",
                    help="time the scalar C restatement when the compiled reference probe (oracle/_ref) is missing, "
                         "instead of failing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    if args.gpus > 1:  # ranks of a node share few host CPUs and the point of the run is RCCL: both default on
        args.device_positions = not args.host_positions
        args.gather = not args.no_gather

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
    # a launcher that narrows every rank's view to its own GPU (HIP_VISIBLE_DEVICES per rank) leaves one visible device
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")  # "nccl" is RCCL over xGMI on ROCm
    group = Group(backend=backend, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
    seen_backend, seen_world = group.backend_world()
    if world > 1:
        assert seen_world == world, (seen_world, world)
        assert seen_backend == backend and (backend == "nccl" or "SPX_BENCH_BACKEND" in os.environ), \
            f"N > 1 runs over RCCL (backend nccl); torch.distributed reports {seen_backend}"

    if args.mode == "incremental":
        return incremental_bench(args, sp, torch, group, rank, local_rank, world)

    # ---- network: rank 0 owns the blob; the other ranks receive it over RCCL ----
    t_net = time.perf_counter()
    if rank == 0 or world == 1:
        blob = np.fromfile(args.net, dtype=np.uint8) if args.net else sp.synthetic_net_bytes(args.preset)
    else:
        blob = None
    blob = group.broadcast_bytes(blob)
    net_bcast_s = time.perf_counter() - t_net
    net = sp.Network(blob)
    state = sp.NnueState(net, device=local_rank, max_batch=args.batch)

    # ---- workload: this rank's shard of seeded random legal positions, resident in HBM ----
    n_distinct = min(args.distinct or args.batch, args.batch)
    if args.device_positions:
        d_distinct = torch.empty((n_distinct, 32), dtype=torch.uint8, device="cuda")
        state.random_positions_device(d_distinct.data_ptr(), n_distinct, seed=20260927 + rank, min_ply=8, max_ply=120, dfrc_every=4)
        distinct = d_distinct.cpu().numpy().reshape(-1).view(sp.PACKED_DTYPE)  # for the oracle / row-count legs
    else:
        distinct = sp.random_positions(n_distinct, seed=20260927 + rank, min_ply=8, max_ply=120, dfrc_every=4)
        d_distinct = torch.from_numpy(distinct.view(np.uint8).reshape(-1, 32)).cuda()
    if n_distinct < args.batch:  # tiled ON THE DEVICE: an HBM-filling batch (config 5) never exists in host memory
        reps = -(-args.batch // n_distinct)
        d_pos = d_distinct.repeat(reps, 1)[: args.batch].contiguous()
        positions = distinct  # the sample checks below look at the first n_sample <= n_distinct positions
    else:
        d_pos, positions = d_distinct, distinct
    pipelined = not args.no_pipeline

    elapsed, (sort_ms, ft_ms, mlp_ms, calls), settle_steps, d_last = timed_full_run(args, torch, group, state, d_pos, pipelined)
    rank_ft_ms = timed_full_run.per_rank_ft_ms
    timed_prepare_ms = timed_full_run.prepare_ms
    # checksum of checksums over all shards (summed in slabs: no 8-byte copy of an HBM-filling score array)
    checksum = group.sum_int(sum(int(d_last[lo:lo + (1 << 26)].sum(dtype=torch.int64).item())
                                 for lo in range(0, args.batch, 1 << 26)))

    # ---- parity inside the bench (outside the timed region): oracle on a sample of every rank's shard ----
    n_sample = min(4096, n_distinct)
    got = d_last[:n_sample].cpu().numpy()
    exact = group.sum_int(int(oracle_sample_check(sp, blob, positions[:n_sample], got))) == world

    # ---- optional result gather over RCCL (not on the data path; outside the timed region) ----
    gathered_ok = None
    if args.gather and (world > 1 or group.dist):  # (world 1 only under SPX_FORCE_DIST: the RCCL smoke run)
        full = group.gather_scores(d_last.cpu().numpy(), args.batch * world)
        lo = rank * args.batch
        mine_ok = np.array_equal(full[lo:lo + args.batch], d_last.cpu().numpy())
        gathered_ok = group.sum_int(int(mine_ok and int(full.sum(dtype=np.int64)) == checksum)) == world

    # ---- second headline: the same run without compact piece-square rows ----
    wide = None
    if not args.no_wide and state.compact_psq_rows:
        wstate = sp.NnueState(net, device=local_rank, max_batch=args.batch, wide_psq_rows=True)
        w_elapsed, (_, w_ft, _, w_calls), _, w_last = timed_full_run(args, torch, group, wstate, d_pos, pipelined)
        w_same = group.sum_int(int(torch.equal(w_last, d_last))) == world
        wide = {"value": world * args.batch * args.steps / w_elapsed, "unit": "evals/s",
                "ms_per_step": w_elapsed / args.steps * 1e3, "ft_kernel_ms": w_ft / max(w_calls, 1),
                "identical_scores": bool(w_same),
                "note": "same timed run on a context created with SPX_CTX_WIDE_PSQ_ROWS: the ONE-KERNEL path (spx_ft_kernel) with every "
                        "piece-square row fetched as its 2 KiB i16 row - the round-1..3 worst case, kept for comparison. What nets "
                        "with wide piece-square rows get on the default path (high-byte planes beside the LDS slab) is the "
                        "`realistic_rows` line"}
        wstate.close()

    secondary = None
    if world == 1 and not args.no_secondary and not args.net:
        secondary = secondary_legs(args, sp, torch, group, state, net, d_pos, positions, pipelined, local_rank)
    elif world > 1 and not args.no_secondary and not args.net:
        # BASELINE configs[3] (self-play sharded over the GPUs) and the incremental leg on the N > 1 line
        secondary = secondary_legs_multi(sp, torch, group, net, local_rank, rank, world, args.preset)

    if rank == 0:
        wide_rows, compact_rows, thr_rows = state.count_rows(distinct)  # host-side count; tiled batches scale the distinct block
        if n_distinct < args.batch:
            wide_rows, compact_rows, thr_rows = (int(v * (args.batch / n_distinct)) for v in (wide_rows, compact_rows, thr_rows))
        psq_rows = wide_rows + compact_rows
        algo_bytes = 2048 * psq_rows + 1024 * thr_rows + 36 * args.batch  # per launch
        requested = 2048 * wide_rows + 1024 * (compact_rows + thr_rows) + 36 * args.batch  # what the loads ask for
        ft_avg_s = ft_ms / max(calls, 1) / 1e3
        value = world * args.batch * args.steps / elapsed
        pmc = load_pmc("full", batch=args.batch, preset=args.preset, net=args.net) if n_distinct == args.batch else None
        sliced = state.takes_sliced_pipeline(min(args.batch, state.scratch_batch), pipelined=pipelined)
        kp = kernel_pmc(pmc, "spx_ftx_gather_kernel" if sliced else "spx_ft_kernel")
        hbm = {
            "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_position": algo_bytes / args.batch,
            "algorithmic_gbs": algo_bytes / ft_avg_s / 1e9, "peak": HBM_PEAK_GBS,
            "algorithmic_over_peak": algo_bytes / ft_avg_s / 1e9 / HBM_PEAK_GBS,
            "note": "cache-resident gather; HBM not binding: the 100 MB of row tables stay in L2 / Infinity Cache, so the "
                    "SURVEY 8(d) algorithmic-byte rate exceeds the HBM peak (algorithmic_over_peak is NOT a roofline "
                    "fraction). traffic_* = measured fabric+HBM bytes per launch (rocprofv3 FETCH_SIZE x2 per the gfx950 "
                    "correction + WRITE_SIZE, committed PMC passes; Infinity-Cache hits are included, so an upper bound "
                    "on HBM bytes), null if this configuration was not profiled",
            "traffic_bytes_per_launch": None, "traffic_gbs": None, "frac": None,
        }
        if kp and "FETCH_SIZE" in kp["counters"] and "WRITE_SIZE" in kp["counters"]:
            tb = (2 * kp["counters"]["FETCH_SIZE"] + kp["counters"]["WRITE_SIZE"]) * 1024
            hbm.update(traffic_bytes_per_launch=tb, traffic_gbs=tb / ft_avg_s / 1e9, frac=tb / ft_avg_s / 1e9 / HBM_PEAK_GBS,
                       frac_of_achievable=tb / ft_avg_s / 1e9 / HBM_ACHIEVABLE_GBS)
        main_kernel = "spx_ftx_gather_kernel" if sliced else "spx_ft_kernel"
        if sliced:
            # the pipeline's gather asks the L2 for the threat / pawn-pair rows and the high-byte planes of the wide piece-square
            # rows (128 B per row and XCD = 1 KiB per row); the piece-square rows' low-byte planes come from the LDS slab
            # (counted by the pack kernel over what the gather really walks: cold threat / pawn-pair rows + the high-byte planes of every
            # piece-square row that does not fit i8 - near-compact rows included, ADVICE r4 - against piece-square + hot rows from LDS)
            walk = state.ftx_walk(0 if pipelined else -1)
            # (the walk is the LAST pass's, the HIP events bracket all passes of a call (65 536 positions each, the last one
            # the rest): everything below is per CALL - the last pass's counts scaled by positions of the call / positions of that pass,
            # counts being proportional to positions)
            passes = max(1, -(-args.batch // 65536))
            call_scale = args.batch / (args.batch - 65536 * (passes - 1))
            if call_scale != 1.0:
                walk = {k: v * call_scale for k, v in walk.items()}
            requested = 1024 * walk["global_rows"] + 36 * args.batch
            lds_served = 1024 * walk["lds_rows"]
        l2_gbs = requested / ft_avg_s / 1e9
        roofline = {
            "kernel": main_kernel, "bound": "l2", "achieved": l2_gbs, "peak": L2_PEAK_GBS, "unit": "GB/s",
            "frac": l2_gbs / L2_PEAK_GBS, "traffic": hbm["traffic_bytes_per_launch"],
            "requested_bytes_per_launch": requested, "ft_kernel_ms": ft_avg_s * 1e3,
            "rows_per_launch": ({"through_the_texture_path_1KiB": walk["global_rows"], "from_lds_1KiB": walk["lds_rows"]} if sliced else
                                {"psq_wide_2KiB": wide_rows, "psq_compact_1KiB": compact_rows, "threat_1KiB": thr_rows}),
            "note": ("achieved = bytes the gather's row loads ask the L2s for (1 KiB per threat / pawn-pair row and per high-byte "
                     "plane of a wide piece-square row, summed over the 8 column slices; + record and score) / the gather "
                     "kernel's HIP-event duration, against the aggregate L2 bandwidth. The piece-square rows (lds_served_bytes) "
                     "come from the king bucket's slab in LDS and are NOT in `achieved`; rows_incl_lds_gbs counts them too, for "
                     "comparison with the one-kernel path's figure (secondary.full_refresh_paths). The committed PMC pass counts "
                     "what the kernel really asked for - padding rows, row lists and slab fills included - as TCC_REQ x 128 B "
                     "(l2_pmc)") if sliced else
                    ("achieved = bytes requested by the kernel's row loads (2 KiB per wide piece-square row, 1 KiB per "
                     "compact piece-square / threat / pawn-pair row, + record and score) / the FT kernel's HIP-event "
                     "duration, against the aggregate L2 bandwidth; the committed PMC pass counts the same bytes as "
                     "TCC_REQ x 128 B (l2_pmc)"),
            "l2_pmc": None, "hbm": hbm, "valu": None,
        }
        if sliced:
            roofline["lds_served_bytes_per_launch"] = lds_served
            roofline["rows_incl_lds_gbs"] = (requested + lds_served) / ft_avg_s / 1e9
            roofline["before_gather_ms"] = timed_prepare_ms
            # Round 5: what BINDS the gather is not a byte rate but instruction slots of three units of a CU - the texture / L1 path
            # (a 16-byte-per-lane wave load holds it ~17.5 cycles whatever its mask or width), the LDS pipe (8.8 cycles per 1 KiB wave
            # read) and the matrix pipe (16 cycles per v_mfma_i32_16x16x64_i8, PMC) -, so the roofline is stated in them. The counts are
            # EXACT: the pack kernel sums what it lays out for the gather to walk (spx_debug_ftx_walk; steps per column slice x 8).
            # (a batch above one pass of the pipeline - 65 536 positions - is walked in passes: the pack kernel's counts are the last
            # pass's, the HIP events bracket all of a call's passes)
            ft_launch_s = ft_avg_s  # (the events' interval: all passes of a call; the walk above is scaled to the call)
            steps_g, steps_l = 8 * walk["cold_steps"] + walk["high_plane_steps_all_slices"], 8 * walk["lds_steps"]
            l1_instr = 4 * steps_g + 8 * walk["stages"] + 2 * 8 * walk["groups"]  # row loads + stage loads + (head load, output store)
            # rows; entries (round 6: one read per PAIR of steps of the LDS / cold sections - 16-bit entries -, one per step of the
            # high-byte sections); stage + output passes
            lds_instr = (4 * steps_l + (8 * (walk["cold_steps"] + walk["lds_steps"]) + 1) // 2 + walk["high_plane_steps_all_slices"]
                         + 2 * 8 * walk["stages"] + 6 * 8 * walk["groups"])
            mfma_instr = 4 * (steps_g + steps_l)
            floors = {"texture_path": l1_instr * L1_CYCLES_PER_WAVE_LOAD / 256 / CLOCK_HZ,
                      "lds": lds_instr * LDS_CYCLES_PER_WAVE_READ / 256 / CLOCK_HZ,
                      "matrix_pipe": mfma_instr * MFMA_CYCLES / N_SIMDS / CLOCK_HZ}
            bound = max(floors, key=floors.get)
            rows_walked = 4 * (steps_g + steps_l)
            units = {
                "texture_path": {"wave_instructions_per_launch": l1_instr, "cycles_each": L1_CYCLES_PER_WAVE_LOAD,
                                 "floor_us": floors["texture_path"] * 1e6, "busy_frac": floors["texture_path"] / ft_launch_s},
                "lds": {"wave_instructions_per_launch": lds_instr, "cycles_each": LDS_CYCLES_PER_WAVE_READ,
                        "floor_us": floors["lds"] * 1e6, "busy_frac": floors["lds"] / ft_launch_s},
                "matrix_pipe": {"v_mfma_i32_16x16x64_i8_per_launch": mfma_instr, "cycles_each": MFMA_CYCLES,
                                "floor_us": floors["matrix_pipe"] * 1e6, "busy_frac": floors["matrix_pipe"] / ft_launch_s},
            }
            # the roofline proper = the unit with the largest floor (1 KiB per wave instruction on either memory unit)
            peaks = {"texture_path": L1_PEAK_GBS * 16.0 / L1_CYCLES_PER_WAVE_LOAD, "lds": LDS_PEAK_GBS * 8.0 / LDS_CYCLES_PER_WAVE_READ}
            instr = {"texture_path": l1_instr, "lds": lds_instr}
            mem_bound = bound if bound in peaks else "texture_path"
            hw_peaks = {"texture_path": L1_PEAK_GBS, "lds": LDS_PEAK_GBS}
            achieved = instr[mem_bound] * 1024 / ft_launch_s / 1e9
            roofline.update({
                #  peak = the HARDWARE figure of the binding unit (64 B / clock / CU through the texture path, 128 B / clock / CU
                # out of LDS); the ceiling this repo measured for the instruction forms the gather uses (17.5 / 8.8 cycles per 1 KiB
                # wave instruction instead of 16 / 8) is stated beside it
                "bound": {"texture_path": "l1", "lds": "lds"}[mem_bound], "achieved": achieved,
                "peak": hw_peaks[mem_bound], "unit": "GB/s", "frac": achieved / hw_peaks[mem_bound],
                "measured_ceiling": peaks[mem_bound], "frac_of_measured_ceiling": floors[mem_bound] / ft_launch_s,
                "binding_unit": bound, "units": units,
                "passes_per_call": passes, "walk": dict(walk, rows_useful=walk["global_rows"] + walk["lds_rows"], rows_walked_per_slice=rows_walked,
                             padding_efficiency=(walk["global_rows"] + walk["lds_rows"]) / max(rows_walked, 1),
                             hot_rows=int(state.hot_rows().size),
                             lds_share_of_rows=walk["lds_rows"] / max(walk["global_rows"] + walk["lds_rows"], 1)),
                "l2": {"achieved": l2_gbs, "peak": L2_PEAK_GBS, "frac": l2_gbs / L2_PEAK_GBS, "requested_bytes_per_launch": requested},
                "note": ("What binds the gather is instruction slots of three units of a CU, so the roofline is stated in them: bound = the unit "
                         "with the largest floor - 'lds' (the LDS pipe: 8.8 cycles per 1 KiB wave read; piece-square slab + hot threat / "
                         "pawn-pair rows, stage entries) or 'l1' (the texture / L1 path: 17.5 cycles per 16-byte-per-lane wave load whatever "
                         "its mask or width, tools/probes/tcp_mask_probe.hip; cold rows and high-byte planes 4 per step, one stage load per 8 "
                         "steps, a head load and an output store per group). achieved = 1 KiB x that unit's wave instructions per launch "
                         "(counted by the pack kernel, all 8 column slices) / the gather's HIP-event duration IN THE TIMED REGION (pipelined "
                         "steps: beside two other batches' preparation), peak = the unit's HARDWARE rate (256 CUs x 64 B / clock through the "
                         "texture path, x 128 B / clock out of LDS, at 2.4 GHz), frac = achieved / peak; measured_ceiling = 256 CUs x 1 KiB / the "
                         "cycles one such wave instruction was measured to hold the unit x 2.4 GHz, frac_of_measured_ceiling = the unit's floor / "
                         "the duration. units = all three incl. the matrix pipe (16 cycles per v_mfma_i32_16x16x64_i8; the count is the row-adding ones - a section with an odd number of steps enters its loop with four more that add zeros: +7 % by the PMC). "
                         "l2 = round 4's figure (bytes the row loads ask the L2s for / aggregate L2 bandwidth): the kernel got faster by "
                         "asking for LESS; hbm = SURVEY 8(d)'s algorithmic bytes and the measured fabric traffic. "
                         "secondary.full_refresh_paths has the kernel alone (stream-ordered)"),
            })
            roofline["pipeline"] = ("spx_ftx_extract_kernel -> spx_ftx_rank_kernel -> spx_ftx_plan_kernel -> spx_ftx_scatter_kernel "
                                    "-> spx_ftx_gather_kernel; before_gather_ms = HIP-event time between the sorts and the gather "
                                    "(stream-ordered steps: the four preparation kernels; pipelined steps: they run on a stream "
                                    "of their own beside the previous batch's gather, this is what is left of the wait)")
        if kp and "TCC_REQ_sum" in kp["counters"]:
            c = kp["counters"]
            req_bytes, miss_bytes = c["TCC_REQ_sum"] * 128, c.get("TCC_MISS_sum", 0) * 128
            roofline["l2_pmc"] = {"tcc_req_bytes_per_launch": req_bytes,
                                  "hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)),
                                  "kernel_us_under_rocprofv3": kp.get("avg_us"),
                                  # a line that misses passes through the L2 arrays twice (the fill, then the read): what the
                                  # L2 itself moves per launch, against the same 34.5 TB/s
                                  "bytes_incl_fills_per_launch": req_bytes + miss_bytes,
                                  "frac_of_l2_peak_incl_fills": (req_bytes + miss_bytes) / ft_avg_s / 1e9 / L2_PEAK_GBS}
        if pmc:
            roofline["valu"] = valu_block(kp, pmc.get("valu_cycles_per_wave_instr", 4))
        if kp and "TCP_TOTAL_CACHE_ACCESSES_sum" in kp["counters"] and "GRBM_GUI_ACTIVE" in kp["counters"]:
            # the unit the counters show busiest: the CU's texture / L1 path takes one 64-byte access per cycle, a 16-byte-per-lane
            # wave load is 16 of them whatever it coalesces to (MI355X_MICROARCH.md: 64 B/clk/CU)
            acc, cyc = kp["counters"]["TCP_TOTAL_CACHE_ACCESSES_sum"], kp["counters"]["GRBM_GUI_ACTIVE"] / 8.0
            roofline["l1_path"] = {"accesses_per_launch": acc, "bytes_per_access": 64, "kernel_cycles": cyc,
                                   "busy_frac": acc / (N_SIMDS / 4 * cyc),
                                   "mfma_busy_frac": kp["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (N_SIMDS * cyc),
                                   "note": "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x kernel cycles) from the committed PMC passes (the "
                                           "kernel alone: rocprofv3 serialises the dispatches it counts)"}
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
            "bit_exact_sample": bool(exact),
            "config": {
                "workload": (f"BASELINE configs[1]: full-refresh NNUE forward on {args.batch} seeded random legal positions "
                             "per GPU (random playouts 8-120 plies, every 4th game DFRC"
                             + (", generated on the device" if args.device_positions else "") + "), bit-exact vs CPU"
                             + (f"; batch tiled from {n_distinct} distinct positions" if n_distinct < args.batch else "")),
                "batch_per_gpu": args.batch,
                "resident_bytes": int(torch.cuda.max_memory_allocated()) if hasattr(torch.cuda, "max_memory_allocated") else None,
                "chunks_per_step": -(-args.batch // state.scratch_batch),
                "net": (f"file {os.path.basename(args.net)} '{net.name}'" if args.net else
                        f"synthetic CBNF '{net.name}' (Stormphrax 8.0.2 shape: (704x16+64368)->1024)x2->(32x2->64->1)x8"),
                "compact_psq_rows": f"{state.compact_psq_rows} of 11264 piece-square rows fit i8 and are served as 1 KiB copies"
                                    + (f"; {state.near_psq_rows} more have <= 32 weights outside i8: 1 KiB copy + exact remainders"
                                       if state.near_psq_rows else ""),
                "parallelism": f"positions sharded over {world} GPU(s), no collective on the data path; "
                               + ("steps issued through the pipelined spx_eval_full_device_async (two internal streams: "
                                  "sorts / MLP of a batch overlap the next batch's feature-transformer kernel)"
                                  if pipelined else "steps issued stream-ordered on one HIP stream per GPU"),
                "ranks": {"backend": seen_backend, "world": seen_world,
                          "ft_kernel_ms_per_rank": rank_ft_ms,
                          "evals_per_sec_per_rank_kernel_bound": {
                              "min": args.batch / (max(rank_ft_ms) / 1e3), "max": args.batch / (min(rank_ft_ms) / 1e3),
                              "mean": args.batch / (sum(rank_ft_ms) / len(rank_ft_ms) / 1e3)},
                          "note": "per-rank figures are each rank's own FT-kernel HIP-event time (rank order): a straggler "
                                  "GPU shows here; `value` uses the MAX-over-ranks wall time"},
                "net_distribution": (f"rank 0's {blob.size} B net image broadcast to the other ranks over "
                                     f"{backend} in {net_bcast_s * 1e3:.1f} ms (init only)" if world > 1 else "single rank"),
                "settle_steps": settle_steps,
                "bit_exact_sample": f"{n_sample} positions of every rank's shard checked against the CPU oracle "
                                    "(oracle/spx_oracle.c) after the timed region",
                "gathered_scores_ok": gathered_ok,
                "checksum": checksum,
                "kernel_ms": {"sort": sort_ms / max(calls, 1), "before_ft": timed_prepare_ms, "ft": ft_ms / max(calls, 1),
                              "mlp": mlp_ms / max(calls, 1)},
                "full_refresh_path": ("column-sliced pipeline (spx_ftx.hip): extraction, counting sort, plan, gather" if sliced
                                      else "one kernel (spx_ft_kernel)"),
            },
            "roofline": roofline,
        }
        if wide:
            line["wide_psq_rows"] = wide
        if secondary:
            realistic = secondary.pop("realistic_rows", None)
            if realistic:
                line["realistic_rows"] = realistic
            line["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1 and not args.net:  # the CPU leg is timed on rank 0 of the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(sp, distinct, blob, args.cpu_seconds, args.allow_port_baseline)
        print(json.dumps(line), flush=True)
    group.close()


if __name__ == "__main__":
    main()
