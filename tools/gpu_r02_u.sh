cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
echo "== device group + ABI"; timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "device_group" 2>&1 | tail -3 | tee $O/group.txt
echo "== parity of the rolling-window variant"; SPX_LIB=$PWD/variants/libspx_roll4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity_roll4.txt
echo "== full-refresh A/B"; timeout 1800 bash tools/gpu_ab.sh 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_ft.txt
