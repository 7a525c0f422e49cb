# round 2, second GPU run: tests, new bench lines, update-kernel A/B (v1 vs ray-walk), PMC of the incremental bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; tail -c 1500 $O/bench_driver_args.json; tail -3 $O/bench_driver_args.err
for v in 0 1; do
  SPX_UPDATE_V1=$v timeout 600 python bench.py --mode incremental --no-cpu-baseline > $O/bench_inc_v1_$v.json 2> $O/bench_inc_v1_$v.err; tail -c 1200 $O/bench_inc_v1_$v.json; tail -3 $O/bench_inc_v1_$v.err
done
timeout 600 python bench.py --mode incremental --no-cpu-baseline > $O/bench_inc_default.json 2>&1; tail -c 600 $O/bench_inc_default.json
for b in 4096 16384 262144; do timeout 600 python bench.py --mode incremental --no-cpu-baseline --batch $b --steps 100 > $O/bench_inc_$b.json 2>&1; tail -c 400 $O/bench_inc_$b.json; done
timeout 1500 bash tools/gpu_pmc_inc.sh r02b > $O/pmc_inc.txt 2>&1; grep -E "update_kernel|ft_kernel" $O/pmc_inc.txt | head -60
python tools/pmc_to_json.py gpurun_out/pmc_inc_r02b $O/r02_pmc_incremental.json --command "python bench.py --mode incremental --steps 40 --warmup 5 --no-cpu-baseline"
