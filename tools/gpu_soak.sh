# determinism / soak checks: self-play twice with the same seed -> the same games; pipelined vs stream-ordered bench checksums
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do python tools/spx_selfplay.py --games 2048 --target 6144 --dfrc --out gpurun_out/soak$i > /dev/null; done
# (the games live on the device: their ORDER in the file is the order in which seats finish; the SET of games is what a seed fixes)
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from _datagen_rules import parse_games
a, b = (sorted((h, m.tobytes(), s.tobytes()) for h, m, s, _ in parse_games(open(f"gpurun_out/soak{i}.0.vf", "rb").read())) for i in (1, 2))
print("selfplay: same set of games" if a == b else "selfplay: DIFFERENT games", len(a), "games")
PY
rm -f gpurun_out/soak*.vf
for mode in "" "--no-pipeline"; do
  python bench.py --steps 2000 --warmup 20 --no-cpu-baseline $mode | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ${mode:-pipelined}: checksum', j['config']['checksum'], '%.3e' % j['value'])"
done
