# round 2, sixth GPU run: all tests, A/B of the gather variants (full + incremental), config-5 bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -12
echo "== full refresh A/B"; timeout 1200 bash tools/gpu_ab.sh 2 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_full.txt
echo "== incremental A/B"; timeout 1200 bash tools/gpu_ab_inc.sh 65536 2>&1 | grep -v amdgpu.ids | tee $O/ab_inc.txt
echo "== config 5 bench"
timeout 900 python bench.py --batch 6800000000 --distinct 131072 --steps 1 --warmup 0 --no-wide --no-cpu-baseline > $O/bench_config5.json 2> $O/config5.err; tail -c 600 $O/bench_config5.json; tail -3 $O/config5.err
