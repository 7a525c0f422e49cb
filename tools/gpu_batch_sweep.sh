# full-refresh throughput vs batch size (pipelined default and stream-ordered): looks for cliffs between the kernel variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 1024 4096 8192 16384 32768 65536 131072 524288; do
  for mode in "" "--no-pipeline"; do
    echo -n "batch $b ${mode:-pipelined}: "
    python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline $mode | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e evals/s  %.1f us/step' % (j['value'], j['ms_per_step']*1e3))"
  done
done
