# longer self-play run with full validation of the output: every recorded move legal (host replay), device replay identical
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/spx_selfplay.py --games 8192 --target 40000 --dfrc --out gpurun_out/soak | cut -c1-330
python - <<PY
import sys, time
sys.path.insert(0, ".")
import numpy as np, stormphrax_amd as sp
blob = open("gpurun_out/soak.0.vf", "rb").read()
t0 = time.time(); host, games = sp.viri_expand(blob); t1 = time.time()
st = sp.NnueState(sp.Network.synthetic("tame"), device=0, max_batch=1 << 20)
dev, g2, bad = st.viri_expand(blob); t2 = time.time()
print("games", games, "positions", len(host), "host replay %.1f s, device %.2f s" % (t1 - t0, t2 - t1), "identical", host.tobytes() == dev.tobytes(), "bad", bad)
# recorded scores == -evaluate_once(next position), seen from WHITE (the reference's convention): the driver's
# incrementally maintained accumulators never drifted
full = st.evaluate_once(host)
ok = 0
idx = 0
lengths = []
off = 0
while off < len(blob):
    off += 32; n = 0
    while blob[off:off+4] != b"\x00\x00\x00\x00": off += 4; n += 1
    off += 4; lengths.append(n)
start = 0; mism = 0; checked = 0
for n in lengths:
    for k in range(start, start + n - 1):
        want = -int(full[k + 1])
        if host["stm_ep"][k] & 0x80: want = -want
        want = 0 if abs(want) <= 2 else max(-32000, min(32000, want))
        mism += int(host["eval"][k]) != want; checked += 1
    start += n
print("score checks", checked, "mismatches", mism)
PY
rm -f gpurun_out/soak.0.vf
