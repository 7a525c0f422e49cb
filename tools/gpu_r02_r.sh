cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
echo "== incremental A/B: update kernel at 5 / 4 waves per SIMD"; timeout 900 bash tools/gpu_ab_inc.sh 65536 2>&1 | grep -v amdgpu.ids | tee $O/ab_upd_waves.txt
for b in 16 32 48 64; do echo -n "incremental SPX_FT_BLOCKS_PER_CU=$b: "; SPX_FT_BLOCKS_PER_CU=$b python bench.py --mode incremental --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), 'update+refresh %.1f us' % (j['roofline']['update_kernel_ms']*1e3))"; done 2>&1 | tee $O/inc_blocks.txt
