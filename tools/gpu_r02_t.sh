cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
for v in mfmaw mfmaws; do echo "== parity $v"; SPX_LIB=$PWD/variants/libspx_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity_$v.txt; done
if grep -q passed $O/parity_mfmaws.txt && ! grep -q failed $O/parity_mfmaws.txt; then rm variants/libspx_mfmaw.so variants/libspx_mfmaw4.so; else rm variants/libspx_mfmaws.so; fi
echo "== full-refresh A/B"; timeout 1500 bash tools/gpu_ab.sh 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_ft.txt
