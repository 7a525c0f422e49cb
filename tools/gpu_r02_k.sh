cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_incremental.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for coop in 1 0; do for r in 1 2; do echo -n "SPX_REFRESH_COOP=$coop: "; SPX_REFRESH_COOP=$coop python bench.py --mode incremental --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), 'update+refresh %.1f us' % (j['roofline']['update_kernel_ms']*1e3), j['config']['bit_exact_vs_full_refresh'])"; done; done 2>&1 | tee $O/coop_ab.txt
for coop in 1 0; do echo -n "stream-ordered SPX_REFRESH_COOP=$coop: "; SPX_REFRESH_COOP=$coop python bench.py --mode incremental --no-pipeline --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'], '%.1f us/ply' % (j['ms_per_step']*1e3), 'update+refresh %.1f us' % (j['roofline']['update_kernel_ms']*1e3), j['config']['bit_exact_vs_full_refresh'])"; done 2>&1 | tee -a $O/coop_ab.txt
echo "== PMC of the round-1 update kernel"
SPX_UPDATE_V1=1 timeout 900 bash tools/gpu_pmc_inc.sh r02k_v1 --no-pipeline > $O/pmc_v1.log 2>&1; cp gpurun_out/pmc_inc_r02k_v1/summary.txt $O/r02_pmc_incremental_round1_kernel.txt; rm -rf gpurun_out/pmc_inc_r02k_v1; grep update_kernel $O/r02_pmc_incremental_round1_kernel.txt | head -30
echo "== bench picks up the committed PMC json"
python bench.py --steps 20 --warmup 5 --no-wide --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], r['frac'], r['valu'], r['l2_pmc'], r['hbm']['frac'], r['traffic'])"
python bench.py --mode incremental --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], r['frac'], r['valu'], r['traffic'])"
