#!/bin/bash
# round 3, lease 2: device-resident self-play + eval-only children (tests, rates), probe re-run with order-independent sinks
mkdir -p gpurun_out
python -m pytest tests/test_gpu_incremental.py tests/test_gpu_configs.py -x -q -k "selfplay or eval_only or bare_bench or two_rank" > gpurun_out/r03_b_pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r03_b_pytest.log
for g in 4096 1024 16384; do
  SPX_SELFPLAY_TRACE=1 python tools/spx_selfplay.py --games $g --target $((g*2)) --dfrc > gpurun_out/r03_b_selfplay_$g.json 2> gpurun_out/r03_b_selfplay_$g.err; echo "selfplay $g rc=$?"
  tail -2 gpurun_out/r03_b_selfplay_$g.err; cut -c1-700 gpurun_out/r03_b_selfplay_$g.json
done
python tools/gpu_gather_ceiling.py --rounds 3 --out gpurun_out/r03_gather_ceiling.json > gpurun_out/r03_gather_ceiling.log 2>&1; echo "ceiling rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_gather_ceiling.json"))
print("same bytes:", d["all_probe_variants_loaded_the_same_bytes"], "ft_us %.1f" % d["ft_kernel_us"])
for r in d["variants"]:
    print("  %-52s %7.1f us  %6.0f GB/s" % (r["name"], r["us_per_launch"], r["requested_gbs"]))
PY
