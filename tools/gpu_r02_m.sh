cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
SPX_SORT_PHASE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for r in 1 2 3; do for ph in 0 1; do for mode in "" "--no-pipeline"; do echo -n "SPX_SORT_PHASE=$ph $mode: "; SPX_SORT_PHASE=$ph python bench.py --no-cpu-baseline --no-wide --steps 100 $mode 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % j['value'], 'ft %.4f' % j['config']['kernel_ms']['ft'], 'sort %.4f' % j['config']['kernel_ms']['sort'], j['bit_exact_sample'], j['config']['checksum'])"; done; done; done 2>&1 | tee $O/phase_ab.txt
