cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
echo "== VALU rate probe (64-bit forms added)"; timeout 300 variants/valu_rate_probe 2>&1 | tee $O/valu_rate_probe.txt
echo "== parity of the combined variant"; SPX_LIB=$PWD/variants/libspx_both.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity_both.txt
echo "== full-refresh A/B"; timeout 1500 bash tools/gpu_ab.sh 3 2>&1 | grep -v amdgpu.ids | tee $O/ab_ft.txt
echo "== incremental with the update kernel's own grid cap"; python bench.py --mode incremental --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee $O/inc.txt
