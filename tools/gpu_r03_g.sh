#!/bin/bash
# round 3, lease 7: A/B of XCD-chunked dealing of update items (self-play and incremental bench), interleaved
mkdir -p gpurun_out
for r in 1 2 3; do for v in base xcd; do
  export SPX_LIB=$PWD/variants/libspx_$v.so
  a=$(python tools/spx_selfplay.py --games 4096 --target 32768 --dfrc | python -c "import json,sys; d=json.load(sys.stdin); print('%.4g %.3f' % (d['value'], d['gpu_call_fraction']))")
  b=$(python tools/spx_selfplay.py --games 16384 --target 32768 --dfrc | python -c "import json,sys; d=json.load(sys.stdin); print('%.4g' % d['value'])")
  c=$(python bench.py --mode incremental --no-cpu-baseline --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g upd %.4f' % (d['value'], d['roofline']['update_kernel_ms']))")
  echo "$v: selfplay4096 $a | selfplay16384 $b | incremental $c"
done; done 2>&1 | tee gpurun_out/r03_g_ab.txt
unset SPX_LIB
python -m pytest tests/test_gpu_incremental.py -x -q > gpurun_out/r03_g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_g_pytest.log
