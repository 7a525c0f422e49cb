# the column-sliced pipeline against the one-kernel path over batch sizes (option ftx = 1 / 0; pipelined and stream-ordered calls)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 12288 16384 24576 32768 49152 65536 131072 262144; do
  for mode in "" "--no-pipeline"; do
    for f in 0 1; do
      SPX_OPTIONS=ftx=$f,ftx_min=8200 python tools/bench_brief.py "batch $b ${mode:-pipelined} ftx=$f" --batch $b --steps 100 --warmup 20 --no-settle $mode
    done
  done
done
