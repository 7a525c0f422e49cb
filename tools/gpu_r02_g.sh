cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_incremental.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
echo "== incremental A/B"; timeout 1200 bash tools/gpu_ab_inc.sh 65536 262144 2>&1 | grep -v amdgpu.ids | tee $O/ab_inc.txt
echo "== full refresh A/B"; timeout 1200 bash tools/gpu_ab.sh 2 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_full.txt
echo "== config 5 bench"
timeout 900 python bench.py --batch 6800000000 --distinct 131072 --steps 1 --warmup 0 --no-wide --no-cpu-baseline > $O/bench_config5.json 2> $O/config5.err; tail -c 700 $O/bench_config5.json; grep -v amdgpu.ids $O/config5.err | tail -3
