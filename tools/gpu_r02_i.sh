cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py tests/test_gpu_native_host.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
echo "== full refresh A/B (pipelined)"; timeout 900 bash tools/gpu_ab.sh 2 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_full.txt
echo "== full refresh A/B (stream-ordered)"; timeout 900 bash tools/gpu_ab.sh 2 --no-wide --no-pipeline 2>&1 | grep -v amdgpu.ids | tee $O/ab_full_strict.txt
echo "== incremental A/B"; timeout 900 bash tools/gpu_ab_inc.sh 65536 4096 2>&1 | grep -v amdgpu.ids | tee $O/ab_inc.txt
