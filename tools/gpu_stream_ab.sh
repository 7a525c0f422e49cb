# A/B of the non-temporal arena accesses threshold (option stream_acc_min) on the incremental bench and on self-play
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in 8192 16384 32768 65536; do
  for t in 0 1000000000; do
    echo -n "games $g stream_min $t: "
    SPX_OPTIONS=stream_acc_min=$t python bench.py --mode incremental --batch $g --steps 200 --warmup 20 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), j['config']['bit_exact_vs_full_refresh'])"
  done
done
for t in 0 1000000000; do
  for g in 4096 16384; do
    echo -n "selfplay $g stream_min $t: "
    SPX_OPTIONS=stream_acc_min=$t python tools/spx_selfplay.py --games $g --target $((2*g)) | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3e' % j['value'])"
  done
done
