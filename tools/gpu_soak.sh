# determinism / soak checks: self-play twice with the same seed -> identical game files; pipelined vs stream-ordered bench checksums
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do python tools/spx_selfplay.py --games 2048 --target 6144 --dfrc --out gpurun_out/soak$i > /dev/null; done
cmp gpurun_out/soak1.0.vf gpurun_out/soak2.0.vf && echo "selfplay: identical output files ($(stat -c %s gpurun_out/soak1.0.vf) bytes)"
rm -f gpurun_out/soak*.vf
for mode in "" "--no-pipeline"; do
  python bench.py --steps 2000 --warmup 20 --no-cpu-baseline $mode | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ${mode:-pipelined}: checksum', j['config']['checksum'], '%.3e' % j['value'])"
done
