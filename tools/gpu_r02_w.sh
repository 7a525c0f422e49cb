cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
echo "== parity (default build: ray-table targets)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity_default.txt
SPX_LIB=$PWD/variants/libspx_ray6.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity_ray6.txt
echo "== full-refresh A/B"; timeout 1800 bash tools/gpu_ab.sh 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_ft.txt
echo "== incremental A/B"; timeout 900 bash tools/gpu_ab_inc.sh 65536 2>&1 | grep -v amdgpu.ids | tee $O/ab_inc.txt
