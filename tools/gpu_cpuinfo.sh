# what the GPU box's host actually offers: logical CPUs, affinity, cgroup quota; and how the reference probe scales
cd ${GRAFT_REPO_ROOT:-/root/repo}
nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
lscpu | grep -i "model name\|^CPU(s)\|Thread\|Core\|Socket" | head
python3 - <<'PY'
import subprocess, sys
sys.path.insert(0, '.')
import stormphrax_amd as sp
pos = sp.random_positions(2048, seed=1)
cmds = "".join(f"add {sp.position_to_fen(p)}\n" for p in pos)
for t in (1, 4, 8, 16, 32, 64, 128, 256):
    out = subprocess.run(["oracle/_ref/sp_ref_probe_tame"], input=cmds + f"bench {t} 3\nquit\n", capture_output=True, text=True).stdout
    print(t, [l for l in out.splitlines() if l.startswith("B ")])
PY
