#!/bin/bash
# round 3, lease 4: full -m gpu suite (realistic preset, chain kernel, applyImmediately, self-play), default bench with secondary legs
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_d_pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r03_d_pytest.log
python bench.py > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r03_d_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_d_bench.json"))
print("value %.4g ms/step %.4f ft %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["ft_kernel_ms"]))
print("ceiling", d["roofline"].get("ceiling"))
print("wide", d.get("wide_psq_rows", {}).get("value"))
print("realistic", json.dumps(d.get("realistic_rows"))[:600])
print("secondary", json.dumps(d.get("secondary"))[:1500])
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
./stormphrax_amd/spx_raweval --preset tame --walk 7 6000 rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1
