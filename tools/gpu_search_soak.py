"""Soak of the live fixed-node search (SPX_SELFPLAY_SEARCH_NODES): a few hundred games at two budgets played by the device-resident
driver at a realistic seat count, every one replayed move by move through the recursive restatement (tests/_search_rules.py) and
datagen's rules (tests/_datagen_rules.py); leaves sampled against the CPU oracle.   python tools/gpu_search_soak.py [games]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import stormphrax_amd as sp  # noqa: E402
from _search_rules import verify_search_file  # noqa: E402
from conftest import Oracle  # noqa: E402

games = int(sys.argv[1]) if len(sys.argv) > 1 else 300
oracle = Oracle()
blob = sp.synthetic_net_bytes("tame")
oracle.use(blob, "tame")
out = []
for budget, seats, max_plies in ((24, 1024, 90), (60, 256, 60)):
    n = games if budget == 24 else max(8, games // 6)
    with sp.NnueState(sp.Network(blob), device=0, max_batch=seats * 64) as st, tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "g.vf")
        stats = st.selfplay(n_games=seats, target_games=n, out_path=path, max_plies=max_plies, dfrc=True, temperature_cp=0, seed=100 + budget,
                            search_nodes=budget)
        t0 = time.perf_counter()
        tally = {}
        checked, expanded, deepest = verify_search_file(sp, st, oracle, open(path, "rb").read(), max_plies, budget, tally)
        assert checked == stats["positions"] and stats["games"] == n
        out.append({"node_budget": budget, "seats": seats, "games": n, "plies_checked": checked, "nodes_restated": expanded,
                    "nodes_expanded_by_the_driver": stats["steps"], "leaf_evals": stats["evals"], "deepest_iteration": deepest,
                    "how_the_games_ended": tally, "replay_seconds": time.perf_counter() - t0})
print(json.dumps({"live_search_soak": out, "every_move_equals_the_restated_search": True}, indent=1))
