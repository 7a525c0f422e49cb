# round 2, third GPU run: A/B of the update-kernel variants, refresh-pass grid sweep, FT kernel with the new decode / activation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_incremental.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== incremental A/B (65536 games)"; timeout 1500 bash tools/gpu_ab_inc.sh 65536 2>&1 | tee $O/ab_inc.txt
echo "== refresh pass grid sweep"
for w in 256 1024 4096 16384 65536; do echo -n "refresh waves $w: "; SPX_REFRESH_WAVES=$w python bench.py --mode incremental --steps 200 --warmup 20 --no-cpu-baseline | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), 'update+refresh %.1f us' % (j['roofline']['update_kernel_ms']*1e3), j['config']['bit_exact_vs_full_refresh'])"; done 2>&1 | tee $O/refresh_sweep.txt
echo "== full refresh A/B"; timeout 900 bash tools/gpu_ab.sh 2 --no-wide 2>&1 | tee $O/ab_full.txt
