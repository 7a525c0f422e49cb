cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for r in 1 2; do for b in 16 32 48 64 96; do for mode in "" "--no-pipeline"; do echo -n "SPX_FT_BLOCKS_PER_CU=$b $mode: "; SPX_FT_BLOCKS_PER_CU=$b python bench.py --no-cpu-baseline --no-wide --steps 100 $mode 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % j['value'], 'ft %.4f' % j['config']['kernel_ms']['ft'])"; done; done; done 2>&1 | tee $O/blocks_ab.txt
