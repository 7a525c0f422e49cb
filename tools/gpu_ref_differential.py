"""Large differential check on the GPU box: the HIP path against the COMPILED REFERENCE (oracle/_ref/sp_ref_probe_tame,
Stormphrax 8.0.2 built from its own sources with the synthetic `tame` net embedded) on N seeded random legal positions -
far more than the committed goldens hold - plus the 32-byte records against its PackedBoard::pack, plus the C restatement
(oracle/libspx_oracle.so) on the wrapping presets.

    gpurun -- 'python tools/gpu_ref_differential.py --positions 1000000 > gpurun_out/ref_differential.json'

Test infrastructure only (like tests/): the product never sees the probe. The positions are generated on the device
(spx_random_positions_gpu), turned into FENs by the host chess core and piped to `--procs` probe processes as `eval <fen>`
lines; the probe answers `E <NnueState::evaluateOnce>` per line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reference_lines(probe, command, prefix, fens, procs):
    """Second token of the probe's `prefix` answer to `command <fen>` for every FEN, by `procs` probe processes."""
    import threading

    bounds = np.linspace(0, len(fens), procs + 1).astype(int)
    out = [None] * len(fens)

    def feed(lo, hi):
        p = subprocess.Popen([probe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        stdout, _ = p.communicate("".join(f"{command} {f}\n" for f in fens[lo:hi]) + "quit\n")
        vals = [ln.split()[1] for ln in stdout.splitlines() if ln.startswith(prefix + " ")]
        assert len(vals) == hi - lo, f"probe answered {len(vals)} of {hi - lo} positions"
        out[lo:hi] = vals

    threads = [threading.Thread(target=feed, args=(lo, hi)) for lo, hi in zip(bounds[:-1], bounds[1:]) if hi > lo]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out


def reference_values(probe, fens, procs):
    """evaluateOnce of every FEN (`eval <fen>` -> `E <value>`), as an int32 array."""
    return np.array([int(v) for v in reference_lines(probe, "eval", "E", fens, procs)], dtype=np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=1000000)
    ap.add_argument("--chunk", type=int, default=262144)
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--port-positions", type=int, default=65536, help="positions per wrapping preset against the C restatement")
    ap.add_argument("--preset", default="tame",
                    help="net preset: its compiled-reference probe oracle/_ref/sp_ref_probe_<preset> must be there (only the tame "
                         "probe is kept in oracle/_ref by default - each embeds 89 MB of net; copy the others from "
                         "oracle/_ref_build/ for a one-off run)")
    ap.add_argument("--no-port", action="store_true", help="skip the C-restatement leg on the wrapping presets")
    ap.add_argument("--pack-positions", type=int, default=1000000,
                    help="leading positions whose 32-byte records are also compared with the reference's PackedBoard::pack")
    args = ap.parse_args()

    import torch

    import stormphrax_amd as sp

    probe = os.path.join(ROOT, "oracle", "_ref", f"sp_ref_probe_{args.preset}")
    assert os.path.exists(probe), f"{probe} is missing (built in the authoring container: make -C oracle ref)"
    report = {"reference": f"compiled Stormphrax 8.0.2 (oracle/_ref/sp_ref_probe_{args.preset}), NnueState::evaluateOnce", "chunks": []}
    net = sp.Network.synthetic(args.preset)
    mismatches, pack_mismatches, packed, done, t0 = 0, 0, 0, 0, time.time()
    distinct_scores = set()
    with sp.NnueState(net, max_batch=args.chunk) as st:
        seed = 90001
        while done < args.positions:
            n = min(args.chunk, args.positions - done)
            d = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
            # every chunk its own seed and ply window; every third game double Fischer random
            st.random_positions_device(d.data_ptr(), n, seed=seed, min_ply=(seed % 5) * 4, max_ply=60 + (seed % 7) * 20, dfrc_every=3)
            torch.cuda.synchronize()
            pos = d.cpu().numpy().reshape(-1).view(sp.PACKED_DTYPE)
            got = st.evaluate_once(pos)
            fens = [sp.position_to_fen(p) for p in pos]
            want = reference_values(probe, fens, args.procs)
            bad = np.nonzero(got != want)[0]
            mismatches += int(bad.size)
            if packed < args.pack_positions:  # wire format (row f-2): the device-written record == PackedBoard::pack(pos, 0)
                hexes = reference_lines(probe, "pack", "K", fens, args.procs)
                pack_mismatches += sum(bytes(p.tobytes()).hex() != h for p, h in zip(pos, hexes))
                packed += n
            distinct_scores.update(np.unique(got).tolist())
            report["chunks"].append({"seed": seed, "positions": n, "mismatches": int(bad.size),
                                     "first_mismatch": None if not bad.size else
                                     {"fen": fens[bad[0]], "hip": int(got[bad[0]]), "reference": int(want[bad[0]])}})
            done += n
            seed += 1
            print(f"[differential] {done}/{args.positions} positions, {mismatches} mismatches, {time.time() - t0:.0f} s", file=sys.stderr)
    report["positions"] = done
    report["mismatches"] = mismatches
    report["packed_board"] = {"positions": packed, "mismatches": pack_mismatches,
                              "what": "records written by spx_random_positions_gpu vs marlinformat::PackedBoard::pack of the same position"}
    report["distinct_scores"] = len(distinct_scores)

    # the wrapping presets have no compiled-reference probe (one binary per embedded net): C restatement, itself pinned on
    # the reference's goldens of those presets
    import ctypes

    so = os.path.join(ROOT, "oracle", "libspx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
    oracle = ctypes.CDLL(so)
    oracle.spxo_init.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    oracle.spxo_eval_mailboxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    report["port"] = {}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest  # the nets with mixed / near-compact piece-square rows are built by the test fixtures

    for preset in (() if args.no_port else ("wild", "extreme", "mixed", "near")):
        blob = (conftest._mixed_rows_net(sp) if preset == "mixed" else conftest._near_rows_net(sp) if preset == "near"
                else sp.synthetic_net_bytes(preset))
        assert oracle.spxo_init(blob.ctypes.data, blob.size) == 0
        pos = sp.random_positions(args.port_positions, seed=4242, min_ply=0, max_ply=200, dfrc_every=3)
        mail, stm = sp.positions_to_mailboxes(pos)
        want = np.empty(len(pos), dtype=np.int32)
        oracle.spxo_eval_mailboxes(mail.ctypes.data, stm.ctypes.data, len(pos), want.ctypes.data)
        with sp.NnueState(sp.Network(blob), max_batch=len(pos)) as st:
            got = st.evaluate_once(pos)
        report["port"][preset] = {"positions": len(pos), "mismatches": int((got != want).sum())}
    report["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(report))
    return 0 if mismatches == 0 and pack_mismatches == 0 and all(v["mismatches"] == 0 for v in report["port"].values()) else 1


if __name__ == "__main__":
    sys.exit(main())
