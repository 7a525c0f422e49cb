# A/B of variants/libspx_*.so on the incremental bench: bash tools/gpu_ab_inc.sh [games...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in ${@:-65536}; do
  for r in 1 2; do
    for l in variants/libspx_*.so; do
      echo -n "$(basename $l) games $g: "
      SPX_LIB=$PWD/$l python bench.py --mode incremental --batch $g --steps 200 --warmup 20 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), j['config']['bit_exact_vs_full_refresh'])"
    done
  done
done
