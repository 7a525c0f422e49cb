# A/B bench of every variants/libspx_*.so (interleaved rounds); usage: bash tools/gpu_ab.sh [rounds] [bench args]
ROUNDS=${1:-3}; shift || true
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - "$ROUNDS" "$@" <<'PY'
import glob, json, os, subprocess, sys
rounds = int(sys.argv[1]); extra = sys.argv[2:]
libs = sorted(glob.glob("variants/libspx_*.so"))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, SPX_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-secondary", "--no-wide"] + extra,
                             env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append((j["value"], j["config"]["kernel_ms"]["ft"], j["config"]["kernel_ms"]["mlp"]))
        except Exception as e:
            res[l].append(("ERR", out.stderr[-300:]))
for l in libs:
    print(os.path.basename(l), " | ".join(f"{v[0]:.3e} ft {v[1]:.4f} mlp {v[2]:.4f}" if v[0] != "ERR" else str(v) for v in res[l]))
PY
