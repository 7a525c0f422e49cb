cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
timeout 3000 bash tools/gpu_final.sh r02 2>&1 | tail -45
