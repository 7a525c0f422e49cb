#!/bin/bash
mkdir -p gpurun_out
for r in 1 2 3; do for v in base stream; do
  export SPX_LIB=$PWD/variants/libspx_$v.so
  c=$(python bench.py --mode incremental --no-cpu-baseline --steps 200 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g upd %.4f' % (d['value'], d['roofline']['update_kernel_ms']))")
  d=$(python bench.py --mode incremental --no-pipeline --no-cpu-baseline --steps 200 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g upd %.4f' % (d['value'], d['roofline']['update_kernel_ms']))")
  e=$(python bench.py --mode incremental --batch 262144 --steps 60 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g' % d['value'])")
  f=$(python tools/gpu_replay_rate.py | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['native_device_ms'])")
  echo "$v: incremental $c | stream-ordered $d | 262144 games $e | config3 replay $f"
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_stream_parents.txt
