#!/bin/bash
# round 3, lease 5: rest of the suite after the realistic-preset threshold fix, group self-play, status kernel; self-play rates
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_e_pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r03_e_pytest.log
for cfg in "4096 8192" "4096 32768" "16384 32768" "1024 2048"; do
  set -- $cfg
  SPX_SELFPLAY_TRACE=1 python tools/spx_selfplay.py --games $1 --target $2 --dfrc > gpurun_out/r03_e_selfplay_$1_$2.json 2> gpurun_out/r03_e_selfplay_$1_$2.err; echo "selfplay $cfg rc=$?"
  grep spx_selfplay gpurun_out/r03_e_selfplay_$1_$2.err; python -c "
import json; d=json.load(open('gpurun_out/r03_e_selfplay_$1_$2.json')); print('  evals/s %.4g gpu_call_fraction %.3f seconds %.3f games %d' % (d['value'], d['gpu_call_fraction'], d['seconds'], d['games']))"
done
