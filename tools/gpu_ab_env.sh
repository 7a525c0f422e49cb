# bench the main library under env-var variants: bash tools/gpu_ab_env.sh "A=1" "B=2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  for r in 1 2; do
    echo -n "$spec: "; env $spec python bench.py --steps 100 --warmup 10 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], j['config']['kernel_ms'], 'frac %.2f' % j['roofline']['frac'])"
  done
done
