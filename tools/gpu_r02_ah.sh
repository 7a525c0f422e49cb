cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ah; mkdir -p $O
for r in 1 2 3 4; do for b in 24 32; do echo -n "selfplay 4096 SPX_UPDATE_BLOCKS_PER_CU=$b: "; SPX_UPDATE_BLOCKS_PER_CU=$b python tools/spx_selfplay.py --games 4096 --target 16384 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'])"; done; done 2>&1 | tee $O/selfplay_blocks.txt
for r in 1 2; do for b in 24 32; do echo -n "selfplay 16384 SPX_UPDATE_BLOCKS_PER_CU=$b: "; SPX_UPDATE_BLOCKS_PER_CU=$b python tools/spx_selfplay.py --games 16384 --target 32768 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % j['value'])"; done; done 2>&1 | tee -a $O/selfplay_blocks.txt
