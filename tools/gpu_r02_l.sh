cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
SPX_FT_POS_MAJOR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for r in 1 2 3; do for pm in 0 1; do for mode in "" "--no-pipeline"; do echo -n "SPX_FT_POS_MAJOR=$pm $mode: "; SPX_FT_POS_MAJOR=$pm python bench.py --no-cpu-baseline --no-wide --steps 100 $mode 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % j['value'], 'ft %.4f' % j['config']['kernel_ms']['ft'], 'sort %.4f' % j['config']['kernel_ms']['sort'], j['bit_exact_sample'], j['config']['checksum'])"; done; done; done 2>&1 | tee $O/posmajor_ab.txt
