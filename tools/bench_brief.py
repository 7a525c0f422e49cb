#!/usr/bin/env python3
"""One short line per bench.py run (A/B helpers): python tools/bench_brief.py [label] [bench args...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1] if len(sys.argv) > 1 else ""
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--no-wide"] + sys.argv[2:],
                     capture_output=True, text=True)
try:
    j = json.loads(out.stdout.strip().splitlines()[-1])
    k = j["config"]["kernel_ms"]
    print(f"{label:28s} {j['value']:.4e} evals/s  {j['ms_per_step']:.4f} ms/step  sort {k['sort']:.4f} ft {k['ft']:.4f} mlp {k['mlp']:.4f}  exact {j['bit_exact_sample']}")
except Exception as exc:  # noqa: BLE001
    print(label, "ERR", exc, out.stderr[-400:])
