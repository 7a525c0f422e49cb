#!/bin/bash
# round 3, lease 3: self-play after removing the per-seat atomics + two plies in flight per half
mkdir -p gpurun_out
python -m pytest tests/test_gpu_incremental.py tests/test_gpu_configs.py -x -q -k "selfplay or eval_only" > gpurun_out/r03_c_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r03_c_pytest.log
for g in 4096 1024 16384; do
  SPX_SELFPLAY_TRACE=1 python tools/spx_selfplay.py --games $g --target $((g*2)) --dfrc > gpurun_out/r03_c_selfplay_$g.json 2> gpurun_out/r03_c_selfplay_$g.err; echo "selfplay $g rc=$?"
  grep spx_selfplay gpurun_out/r03_c_selfplay_$g.err; cut -c1-330 gpurun_out/r03_c_selfplay_$g.json
done
SPX_SELFPLAY_TRACE=1 python tools/spx_selfplay.py --games 4096 --target 32768 --dfrc > gpurun_out/r03_c_selfplay_4096_long.json 2> gpurun_out/r03_c_selfplay_4096_long.err; grep spx_selfplay gpurun_out/r03_c_selfplay_4096_long.err; cut -c1-330 gpurun_out/r03_c_selfplay_4096_long.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03_c_prof -o sp -- python /root/repo/tools/spx_selfplay.py --games 4096 --target 8192 --dfrc > /dev/null 2>&1; cd /root/repo
f=$(ls gpurun_out/r03_c_prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
