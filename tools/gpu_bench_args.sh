# bench with several argument sets: bash tools/gpu_bench_args.sh "--distinct 1" "--distinct 256" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  echo -n "$spec: "; python bench.py --steps 100 --warmup 10 --no-cpu-baseline $spec 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], j['config']['kernel_ms'], 'B/pos %.0f' % j['roofline']['bytes_per_position'], 'frac %.2f' % j['roofline']['frac'])"
done
