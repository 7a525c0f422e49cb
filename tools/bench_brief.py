#!/usr/bin/env python3
"""One short line per bench.py run (A/B helpers): python tools/bench_brief.py [label] [bench args...]  RUNS bench.py with those
arguments; python tools/bench_brief.py FILE (an existing file, or - for stdin) only PRINTS the bench line stored there. (Round 6: piping a
bench run into `bench_brief.py /dev/stdin` started a SECOND, default run beside the first - two processes sharing the GPU - and cost an
evening of chasing a "slow state of the boxes".)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1] if len(sys.argv) > 1 else ""
if label == "-" or (label and os.path.exists(label)):
    text = sys.stdin.read() if label in ("-", "/dev/stdin") else open(label).read()
    j = json.loads([ln for ln in text.strip().splitlines() if ln.startswith("{")][-1])
    k = j.get("config", {}).get("kernel_ms") or {"sort": 0.0, "ft": 0.0, "mlp": 0.0}
    print(f"{label:28s} {j['value']:.4e} {j.get('unit', '')}  {j['ms_per_step']:.4f} ms/step  sort {k['sort']:.4f} ft {k['ft']:.4f} mlp {k['mlp']:.4f}  "
          f"exact {j.get('bit_exact_sample', j.get('config', {}).get('bit_exact_sample'))}")
    raise SystemExit(0)
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--no-wide"] + sys.argv[2:],
                     capture_output=True, text=True)
try:
    j = json.loads(out.stdout.strip().splitlines()[-1])
    k = j["config"]["kernel_ms"]
    print(f"{label:28s} {j['value']:.4e} evals/s  {j['ms_per_step']:.4f} ms/step  sort {k['sort']:.4f} ft {k['ft']:.4f} mlp {k['mlp']:.4f}  exact {j['bit_exact_sample']}")
except Exception as exc:  # noqa: BLE001
    print(label, "ERR", exc, out.stderr[-400:])
