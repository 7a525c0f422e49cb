cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
echo "== pipelined"; timeout 1200 bash tools/gpu_ab.sh 2 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_dyn.txt
echo "== stream-ordered"; timeout 1200 bash tools/gpu_ab.sh 2 --no-wide --no-pipeline 2>&1 | grep -v amdgpu.ids | tee $O/ab_dyn_strict.txt
