cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
echo "== parity sc1"; SPX_LIB=$PWD/variants/libspx_sc1.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee $O/parity_sc1.txt
echo "== full-refresh A/B"; timeout 1800 bash tools/gpu_ab.sh 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_ft.txt
