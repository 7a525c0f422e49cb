# A/B of the split-perspective update kernel (option update_split_max) and the shared-tile MLP (option mlp_share_max):
# incremental ply time vs games. usage: bash tools/gpu_small_ab.sh "<games...>" "<split share>" ["<split share>" ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
GAMES=$1; shift
for g in $GAMES; do
  for cfg in "$@"; do
    split=${cfg% *}; share=${cfg#* }
    echo -n "games $g split<=$split share<=$share: "
    SPX_OPTIONS=update_split_max=$split,mlp_share_max=$share python bench.py --mode incremental --batch $g --steps 200 --warmup 20 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3))"
  done
done
