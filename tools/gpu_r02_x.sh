cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
echo "== parity with the (king bucket, output bucket) order"; SPX_SORT_PHASE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee $O/parity_phase.txt
for r in 1 2 3; do for v in 0 1; do for extra in "" "--no-pipeline"; do echo -n "SPX_SORT_PHASE=$v $extra: "; SPX_SORT_PHASE=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-wide $extra 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % j['value'], 'ft %.4f' % j['config']['kernel_ms']['ft'], 'sort %.4f' % j['config']['kernel_ms']['sort'], j['bit_exact_sample'], j['config']['checksum'])"; done; done; done 2>&1 | tee $O/sort_phase.txt
