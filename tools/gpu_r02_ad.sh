cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ad; mkdir -p $O
echo "== parity incl. near-compact and ten-bit rows"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity.txt
echo "== rates"; timeout 1200 python tools/gpu_near_rate.py 2>$O/near_rate.err | tee $O/near_rate.json | python -c "import json,sys; [print(c) for c in json.loads(sys.stdin.read())['cases']]"; tail -3 $O/near_rate.err
echo "== headline unchanged?"; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-wide 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % j['value'], j['config']['kernel_ms'])"
