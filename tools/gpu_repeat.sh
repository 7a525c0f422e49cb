# repeatability of the bench line over library builds: bash tools/gpu_repeat.sh <rounds> "<bench args>" lib-or-"-" ...
R=$1; shift; ARGS=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 $R); do for l in "$@"; do
  if [ "$l" = "-" ]; then unset SPX_LIB; else export SPX_LIB=$PWD/$l; fi
  echo -n "$l $ARGS: "; python bench.py --no-secondary --no-cpu-baseline --no-wide $ARGS | python tools/bench_brief.py /dev/stdin
done; done
