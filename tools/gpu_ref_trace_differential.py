"""Large differential check of the INCREMENTAL path on the GPU box: the compiled reference (oracle/_ref/sp_ref_probe_tame)
walks random make/unmake trees from many random roots through its own NnueState::push / pop / evaluate (lazy multi-ply
updates, finny-table refreshes and all - oracle/ref_probe.cpp `trace`), the recorded trees are replayed through
spx_acc_replay_tree (update kernels + rebuild passes + MLP) and every EVAL must equal the reference's.

    gpurun -- 'python tools/gpu_ref_trace_differential.py --roots 64 --evals 16384 > gpurun_out/ref_trace_differential.json'

Test infrastructure only, like tests/ (the committed goldens hold 5 such traces; this runs hundreds of times more nodes).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--roots", type=int, default=64)
    ap.add_argument("--evals", type=int, default=16384)
    ap.add_argument("--depth", type=int, default=14)
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--preset", default="tame", help="net preset; needs oracle/_ref/sp_ref_probe_<preset> (see gpu_ref_differential.py)")
    args = ap.parse_args()

    import stormphrax_amd as sp
    from stormphrax_amd import trace as tr

    probe = os.path.join(ROOT, "oracle", "_ref", f"sp_ref_probe_{args.preset}")
    assert os.path.exists(probe), f"{probe} is missing (built in the authoring container: make -C oracle ref)"
    roots = sp.random_positions(args.roots, seed=777, min_ply=0, max_ply=140, dfrc_every=3)
    fens = [sp.position_to_fen(p) for p in roots]
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="spx_traces_")
    # the reference records the trees: `procs` probes at a time
    paths, running = [], []
    for k, fen in enumerate(fens):
        path = os.path.join(tmp, f"trace_{k}.txt")
        paths.append(path)
        p = subprocess.Popen([probe], stdin=subprocess.PIPE, stdout=open(path, "w"), text=True)
        p.stdin.write(f"trace {1000 + k} {args.evals} {args.depth} {fen}\nquit\n")
        p.stdin.close()
        running.append(p)
        if len(running) >= args.procs:
            running.pop(0).wait()
    for p in running:
        p.wait()
    recorded_s = time.time() - t0

    net = sp.Network.synthetic(args.preset)
    report = {"reference": f"compiled Stormphrax 8.0.2 (oracle/_ref/sp_ref_probe_{args.preset}): NnueState::reset/push/pop/evaluate on random DFS walks",
              "roots": args.roots, "depth": args.depth, "traces": []}
    total_evals = total_updates = mismatches = once_mismatches = 0
    gpu_ms = 0.0
    with sp.NnueState(net, max_batch=65536) as st:
        for k, path in enumerate(paths):
            t = tr.Trace(path)
            if t.root_fen is None or not t.evals:
                report["traces"].append({"root": fens[k], "error": "empty trace"})
                mismatches += 1
                continue
            got, want, ms = tr.replay_native(st, t)
            once = st.evaluate_once(t.positions()[np.array([e[0] for e in t.evals], dtype=np.int64)])
            want_once = np.array([e[2] for e in t.evals], dtype=np.int32)
            bad = int((got != want).sum())
            bad_once = int((once != want_once).sum())
            mismatches += bad
            once_mismatches += bad_once
            total_evals += len(t.evals)
            total_updates += t.n_nodes - 1
            gpu_ms += ms
            report["traces"].append({"root": fens[k], "nodes": t.n_nodes, "evals": len(t.evals), "mismatches": bad,
                                     "evaluate_once_mismatches": bad_once, "device_ms": round(ms, 3)})
            os.remove(path)
    report.update({"updates": total_updates, "evals": total_evals, "mismatches": mismatches,
                   "evaluate_once_mismatches": once_mismatches, "device_ms_total": round(gpu_ms, 2),
                   "reference_recording_s": round(recorded_s, 1), "seconds": round(time.time() - t0, 1)})
    print(json.dumps(report))
    return 0 if mismatches == 0 and once_mismatches == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
