cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
import sys
sys.path.insert(0, ".")
import stormphrax_amd as sp
blob = b"".join(sp.viri_random_game(7000 + s, plies=160, dfrc=(s % 4 == 0)) for s in range(2000))
open("gpurun_out/games.vf", "wb").write(blob)
print(len(blob))
PY
time python tools/spx_rescore.py gpurun_out/games.vf gpurun_out/a.bin
time python tools/spx_rescore.py gpurun_out/games.vf gpurun_out/b.bin --validate
cmp gpurun_out/a.bin gpurun_out/b.bin && echo IDENTICAL
rm -f gpurun_out/games.vf gpurun_out/a.bin gpurun_out/b.bin
