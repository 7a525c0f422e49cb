# round-2 opening measurement: VALU issue-rate probe, GPU tests, default bench, PMC of the incremental kernels
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
rocminfo | grep -m3 -E 'gfx|Compute Unit'
timeout 120 tools/probes/valu_rate_probe > gpurun_out/r02a/valu_rate_probe.txt 2>&1; cat gpurun_out/r02a/valu_rate_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver_args.json 2> gpurun_out/r02a/bench_driver_args.err; tail -c 600 gpurun_out/r02a/bench_driver_args.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2>&1; tail -c 300 gpurun_out/r02a/bench_default.json
timeout 600 python bench.py --mode incremental --no-cpu-baseline > gpurun_out/r02a/bench_inc.json 2>&1; tail -c 900 gpurun_out/r02a/bench_inc.json
timeout 1500 bash tools/gpu_pmc_inc.sh r02a > gpurun_out/r02a/pmc_inc.txt 2>&1; tail -60 gpurun_out/r02a/pmc_inc.txt
