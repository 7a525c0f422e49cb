cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ae; mkdir -p $O
echo "== parity (all presets go through the MLP)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | tail -2 | tee $O/parity.txt
echo "== full-refresh A/B (pipelined)"; timeout 900 bash tools/gpu_ab.sh 3 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_quad.txt
echo "== full-refresh A/B (stream-ordered)"; timeout 900 bash tools/gpu_ab.sh 2 --no-wide --no-pipeline 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_quad.txt
echo "== incremental A/B"; timeout 900 bash tools/gpu_ab_inc.sh 65536 2>&1 | grep -v amdgpu.ids | tee $O/ab_inc.txt
