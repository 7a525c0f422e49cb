# Interleaved A/B over (library, SPX_OPTIONS) pairs: bash tools/gpu_ab2.sh <rounds> "<bench args>" "name|variants/libspx_x.so or -|options or -" ...
ROUNDS=${1:-2}; shift; ARGS=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - "$ROUNDS" "$ARGS" "$@" <<'PY'
import json, os, subprocess, sys
rounds = int(sys.argv[1]); extra = sys.argv[2].split(); specs = [s.split("|") for s in sys.argv[3:]]
res = {s[0]: [] for s in specs}
for r in range(rounds):
    for name, lib, opts in specs:
        env = dict(os.environ)
        if lib != "-": env["SPX_LIB"] = os.path.abspath(lib)
        if opts != "-": env["SPX_OPTIONS"] = opts
        out = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-secondary", "--no-wide"] + extra,
                             env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            k = j["config"]["kernel_ms"]
            res[name].append("%.4e ft %.4f mlp %.4f real %.3e" % (j["value"], k["ft"], k["mlp"], j.get("realistic_rows", {}).get("value", 0)))
        except Exception as e:
            res[name].append("ERR " + out.stderr[-300:].replace("\n", " "))
for name, _, _ in specs:
    print("%-28s" % name, " | ".join(res[name]))
PY
