# longer self-play run with FULL validation of the output (tests/_datagen_rules.py: every move legal, device replay
# identical, oracle sample, verification filter, every game replayed through the restated datagen rules)
# usage: bash tools/gpu_selfplay_soak.sh [target games] [net preset]
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SOAK_PRESET=${2:-tame}
python tools/spx_selfplay.py --games 8192 --target ${1:-40000} --dfrc --max-plies 300 --preset $SOAK_PRESET --out gpurun_out/soak | cut -c1-330
python - <<PY
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, stormphrax_amd as sp
from conftest import Oracle
from _datagen_rules import verify_selfplay_file
blob = open("gpurun_out/soak.0.vf", "rb").read()
preset = os.environ["SOAK_PRESET"]
net = sp.synthetic_net_bytes(preset)
oracle = Oracle(); oracle.use(net, preset)
st = sp.NnueState(sp.Network(net), device=0, max_batch=1 << 20)
t0 = time.time()
tally = {}
checked = verify_selfplay_file(sp, st, oracle, blob, max_plies=300, oracle_sample=65536, tally=tally)
print("soak (net preset %s): %d plies of %d bytes verified in %.0f s: legal moves, device replay identical, 65 536 positions vs the oracle, "
      "verification filter, end ply / outcome byte / scores of every game per the restated datagen rules" % (preset, checked, len(blob), time.time() - t0))
print("how the games ended: " + "; ".join("%s %d" % kv for kv in sorted(tally.items(), key=lambda kv: -kv[1])))
PY
rm -f gpurun_out/soak.0.vf
