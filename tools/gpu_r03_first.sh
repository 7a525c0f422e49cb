#!/bin/bash
# round 3, first lease: the whole -m gpu suite on the pruned kernels, the gather-ceiling probe, the default bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_first_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_first_pytest.log
tail -5 gpurun_out/r03_first_pytest.log
python tools/gpu_gather_ceiling.py --out gpurun_out/r03_gather_ceiling.json > gpurun_out/r03_gather_ceiling.log 2>&1; echo "ceiling rc=$?"
python tools/gpu_gather_ceiling.py --wide --out gpurun_out/r03_gather_ceiling_wide.json >> gpurun_out/r03_gather_ceiling.log 2>&1; echo "ceiling wide rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_gather_ceiling.json", "gpurun_out/r03_gather_ceiling_wide.json"):
    try:
        d = json.load(open(f))
        print(f, "same bytes:", d["all_probe_variants_loaded_the_same_bytes"], "ft_us %.1f" % d["ft_kernel_us"])
        for r in d["variants"]:
            print("  %-52s %7.1f us  %6.0f GB/s" % (r["name"], r["us_per_launch"], r["requested_gbs"]))
    except Exception as e:
        print(f, "failed:", e)
PY
python bench.py --no-secondary > gpurun_out/r03_first_bench.json 2> gpurun_out/r03_first_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r03_first_bench.json
