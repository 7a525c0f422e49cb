set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -m3 -E 'gfx|Compute Unit' 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -5
