cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
echo "== full refresh A/B"; timeout 1500 bash tools/gpu_ab.sh 3 --no-wide 2>&1 | grep -v amdgpu.ids | tee $O/ab_full.txt
