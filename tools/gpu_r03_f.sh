#!/bin/bash
# round 3, lease 6: A/B of the FT-kernel prefetch variants; the new search-trace test + bench secondary legs
mkdir -p gpurun_out
bash tools/gpu_ab.sh 3 > gpurun_out/r03_f_ab.txt 2>&1; cat gpurun_out/r03_f_ab.txt
python -m pytest tests/test_gpu_incremental.py tests/test_gpu_parity.py -x -q -k "alpha_beta or golden_vectors" > gpurun_out/r03_f_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_f_pytest.log
