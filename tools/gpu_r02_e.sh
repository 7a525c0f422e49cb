# round 2, fifth GPU run: new tests, config-5 bench, PMC passes (full refresh + incremental) for the roofline JSONs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_configs.py tests/test_gpu_wire.py tests/test_gpu_incremental.py -m gpu -x -q -s 2>&1 | tail -12
echo "== config 5 bench"
timeout 900 python bench.py --batch 6800000000 --distinct 131072 --steps 1 --warmup 0 --no-wide --no-cpu-baseline > $O/bench_config5.json 2> $O/config5.err; tail -c 1500 $O/bench_config5.json; tail -3 $O/config5.err
echo "== PMC full refresh"
timeout 1500 bash tools/gpu_profile.sh r02e > $O/profile_full.log 2>&1; cp gpurun_out/prof_r02e/summary.txt $O/rocprofv3_summary_full.txt; cp gpurun_out/prof_r02e/pmc.json $O/r02_pmc_full_refresh.json; grep -E "ft_kernel" $O/rocprofv3_summary_full.txt | head -40
echo "== PMC incremental"
timeout 1500 bash tools/gpu_pmc_inc.sh r02e > $O/profile_inc.log 2>&1; cp gpurun_out/pmc_inc_r02e/summary.txt $O/rocprofv3_summary_incremental.txt; cp gpurun_out/pmc_inc_r02e/pmc.json $O/r02_pmc_incremental.json; grep -E "update_kernel|ft_kernel" $O/rocprofv3_summary_incremental.txt | head -50
rm -rf gpurun_out/prof_r02e gpurun_out/pmc_inc_r02e
