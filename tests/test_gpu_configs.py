"""BASELINE.json configurations at (or near) their stated sizes on one MI355X: configs[3] self-play with 4 096 concurrent
games, configs[4] the HBM-filling batch, and the N = 2 control flow of bench.py / tools/spx_selfplay.py with both ranks
on this box's one GPU (the driver measures real multi-GPU scaling itself)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def game_lengths(blob):
    lengths, off = [], 0
    while off < len(blob):
        off += 32
        n = 0
        while blob[off:off + 4] != b"\x00\x00\x00\x00":
            off += 4
            n += 1
        off += 4
        lengths.append(n)
    return lengths


def test_config3_selfplay_with_4096_concurrent_games(sp, net_blob, oracle, tmp_path):
    """configs[3] at its stated width: 4 096 concurrent games (double-Chess960 starts) living on the device - moves generated,
    ~35 siblings per seat evaluated WITHOUT being stored (eval-only children), the move chosen, the game records kept and the
    reference's datagen rules applied by spx_game_step_kernel; the host reads counters. Checked against the file
    (tests/_datagen_rules.py: a plain-Python restatement of datagen.cpp:176-300): every move legal, device replay identical,
    4 096 sampled positions equal the CPU ORACLE, every opening passes the verification filter, every game ends at the ply
    and with the outcome byte and scores the reference's loop produces."""
    from _datagen_rules import verify_selfplay_file

    st = sp.NnueState(sp.Network(net_blob("tame")), device=0, max_batch=4096 * 64)
    try:
        path = str(tmp_path / "games.vf")
        stats = st.selfplay(n_games=4096, target_games=5000, out_path=path, max_plies=200, dfrc=True, temperature_cp=20, seed=4)
        assert stats["games"] == 5000 and sum(stats["outcomes"]) == 5000 and stats["evals"] > 5_000_000
        blob = open(path, "rb").read()
        oracle.use(net_blob("tame"), "tame")
        checked = verify_selfplay_file(sp, st, oracle, blob, max_plies=200, oracle_sample=4096)
        assert checked == stats["positions"] and checked > 300_000
        print("host waiting for the GPU %.0f %% of %.3f s" % (100 * stats["gpu_seconds"] / stats["seconds"], stats["seconds"]))
    finally:
        st.close()


def test_config4_hbm_filling_batch():
    """configs[4]: the largest position batch that fits this GPU's HBM (36 bytes resident per position: record in, score
    out; intermediates only for one 4 Mi chunk per lane), evaluated in ONE call - tests/_config5_worker.py, run as its own
    process because it needs torch for a quarter-terabyte device buffer (torch must initialise its HIP runtime before
    libspx_nnue.so is loaded; this pytest process did it the other way round)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_config5_worker.py")], cwd=ROOT, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print(line)
    assert line["tiles_identical"] and line["oracle_sample_exact"] and line["chunks"] > 100
    assert line["in_use_gb"] > 0.8 * line["total_gb"] and line["in_use_gb"] > 200


@pytest.mark.parametrize("script", ["bench", "selfplay"])
def test_two_rank_control_flow_on_one_gpu(script, tmp_path):
    """The N > 1 paths of bench.py (net broadcast, barrier / MAX / SUM, score gather) and tools/spx_selfplay.py under
    torch.distributed.run with 2 ranks sharing GPU 0 (gloo for the collectives: RCCL refuses two ranks on one device)."""
    env = dict(os.environ, SPX_BENCH_SHARE_GPU="1", SPX_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                "127.0.0.1", "--master-port", str(free_port())]
    if script == "bench":
        cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "16384",
                          "--gather", "--no-wide"]
    else:
        cmd = launcher + [os.path.join(ROOT, "tools", "spx_selfplay.py"), "--games", "512", "--target", "700", "--dfrc",
                          "--out", str(tmp_path / "sp")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2
    if script == "bench":
        assert line["bit_exact_sample"] is True and line["config"]["gathered_scores_ok"] is True
        assert line["value"] > 1e6 and "cpu_baseline" not in line
        # the N > 1 line carries BASELINE configs[3] (self-play sharded over the ranks) and the incremental leg
        sec = line["secondary"]
        c4 = sec["config4_selfplay"]
        assert c4["n_gpus"] == 2 and c4["value"] > 1e6 and c4["concurrent_games"] == 4096 and c4["concurrent_games_per_gpu"] == 2048
        assert len(c4["leaf_evals_per_sec_per_rank"]) == 2 and c4["games"] >= 32768
        assert sec["config4_selfplay_4096_games_per_gpu"]["concurrent_games"] == 8192
        inc = sec["incremental"]
        assert inc["n_gpus"] == 2 and inc["bit_exact_vs_full_refresh"] is True and len(inc["updates_plus_evals_per_sec_per_rank"]) == 2
    else:
        assert line["games"] == 1400 and sum(line["outcomes_white_loss_draw_win"]) == 1400
        assert os.path.getsize(tmp_path / "sp.0.vf") > 0 and os.path.getsize(tmp_path / "sp.1.vf") > 0


def test_bare_bench_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE (the shape of the driver's N = 1 command) starts its
    two ranks itself; N > 1 defaults to device-generated positions and the score gather."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPX_BENCH_SHARE_GPU="1", SPX_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--batch", "16384", "--no-wide", "--no-secondary"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "launching 2 ranks" in out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["bit_exact_sample"] is True and line["config"]["gathered_scores_ok"] is True
    assert "generated on the device" in line["config"]["workload"]
    ranks = line["config"]["ranks"]
    assert ranks["world"] == 2 and len(ranks["ft_kernel_ms_per_rank"]) == 2


def test_every_collective_once_through_rccl(tmp_path):
    """No multi-GPU node has been available to any round, and the 2-rank tests share one GPU over gloo: RCCL itself would
    otherwise meet this code for the first time on the driver's 8-GPU lease. SPX_FORCE_DIST=1 initialises the process group
    at world size 1, so every collective the N > 1 runs use - the net broadcast (89 MB, uint8), MAX / SUM all_reduce (f64,
    i64), the per-rank all_gather (f64) and the score all_gather (i32), barriers - goes through the real RCCL library once,
    on device tensors, in bench.py and in the self-play tool."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPX_FORCE_DIST="1", MASTER_PORT="29641")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--gather",
                          "--device-positions", "--no-wide", "--no-secondary", "--no-cpu-baseline"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["ranks"]["backend"] == "nccl" and line["config"]["ranks"]["world"] == 1
    assert line["bit_exact_sample"] is True and line["config"]["gathered_scores_ok"] is True
    env["MASTER_PORT"] = "29642"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spx_selfplay.py"), "--games", "256", "--target", "300",
                          "--out", str(tmp_path / "g")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["collectives"] == "nccl" and line["games"] == 300 and os.path.getsize(str(tmp_path / "g.0.vf")) > 0


def test_device_group_shards_a_batch_over_its_members(sp, net_blob, oracle):
    """spx_group (the C ABI's multi-device entry): two members on this box's one GPU evaluate contiguous shards on their own
    host threads - scores identical to one context over the whole batch and to the CPU oracle, for ragged and tiny
    batches; the default group has one member per visible device."""
    net = sp.Network(net_blob("tame"))
    pos = sp.random_positions(20001, seed=77)
    with sp.NnueState(net, max_batch=32768) as st:
        want = st.evaluate_once(pos)
        want_adj = st.adjust(pos, want, contempt=(3, -5))
    oracle.use(net_blob("tame"), "tame")
    mail, stm = sp.positions_to_mailboxes(pos[:2000])
    assert np.array_equal(want[:2000], oracle.eval_mailboxes(mail, stm))
    with sp.DeviceGroup(net, devices=[0, 0], max_batch_per_device=16384) as grp:
        assert len(grp) == 2 and grp.shard(20001, 0) == (0, 10001) and grp.shard(20001, 1) == (10001, 20001)
        got = grp.evaluate_once(pos)
        assert np.array_equal(got, want)
        assert np.array_equal(grp.adjust(pos, got, contempt=(3, -5)), want_adj)
        for n in (1, 2, 3, 4097):  # one member idle / odd splits
            assert np.array_equal(grp.evaluate_once(pos[:n]), want[:n])
        # more than both members hold at once in one call: each member walks its shard in chunks
        big = np.concatenate([pos, pos])
        assert np.array_equal(grp.evaluate_once(big), np.concatenate([want, want]))
    with sp.DeviceGroup(net, max_batch_per_device=4096) as grp:
        assert len(grp) == sp.device_count() >= 1
        assert np.array_equal(grp.evaluate_once(pos[:5000]), want[:5000])


def test_device_group_plays_its_games_on_every_member(sp, net_blob, oracle, tmp_path):
    """spx_group_selfplay_run (configs[3] from one native process): two members on this box's one GPU play their shares of
    the games concurrently on their own host threads; every member's file obeys the datagen rules and the totals add up."""
    from _datagen_rules import verify_selfplay_file

    net = sp.Network(net_blob("tame"))
    oracle.use(net_blob("tame"), "tame")
    with sp.DeviceGroup(net, devices=[0, 0], max_batch_per_device=16384) as grp:
        stats = grp.selfplay(n_games=257, target_games=401, out_path=str(tmp_path / "g"), max_plies=150, dfrc=True,
                             temperature_cp=15, seed=21)
    assert stats["games"] == 401 and sum(stats["outcomes"]) == 401
    with sp.NnueState(net, max_batch=16384) as st:
        checked = 0
        for member, games in ((0, 201), (1, 200)):
            blob = open(tmp_path / f"g.{member}.vf", "rb").read()
            positions, n = sp.viri_expand(blob)
            assert n == games
            checked += verify_selfplay_file(sp, st, oracle, blob, max_plies=150, oracle_sample=512)
    assert checked == stats["positions"]
    # fewer seats than members: the first members get one seat each and share the whole target; the third member sits out
    with sp.DeviceGroup(net, devices=[0, 0, 0], max_batch_per_device=4096) as grp:
        stats = grp.selfplay(n_games=2, target_games=5, out_path=str(tmp_path / "few"), max_plies=60, seed=3)
    assert stats["games"] == 5 and os.path.exists(tmp_path / "few.0.vf") and os.path.exists(tmp_path / "few.1.vf")
    assert not os.path.exists(tmp_path / "few.2.vf")
