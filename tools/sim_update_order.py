"""would WALKING a batch of independent updates in another order lift the update kernel's L2 hit rate?
CPU only: the host emulation of the kernel's delta derivation (spx_debug_delta) on the bench generator's positions, one random legal
move each; per XCD an LRU model of its L2 (rows of 1 KiB; the walk deals chunks of 64 neighbouring records to the 8 XCDs in turn, as
ItemWalk does). Measured hit rate of the unsorted kernel: 40 % (profiles/r05_pmc_incremental.json) - the model says 32-39 %.

    python tools/sim_update_order.py [records] > profiles/r06_update_walk_order_lru_model.txt
"""
import sys, os, time
import numpy as np
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormphrax_amd as sp
N=int(sys.argv[1]) if len(sys.argv)>1 else 20000
rng=np.random.default_rng(5)
pos=sp.random_positions(N, seed=20260927, min_ply=8, max_ply=120, dfrc_every=4)
mail,stm=sp.positions_to_mailboxes(pos)
recs=[]
for i in range(len(pos)):
    words,kids,_=sp.legal_moves(pos[i])
    if len(words)==0: continue
    j=int(rng.integers(len(words)))
    child=kids[j]; w=int(words[j]); fr=w&63; to=(w>>6)&63
    piece=int(mail[i][fr])
    rows=[]
    for c in (0,1):
        d=sp.debug_delta(pos[i],child,c)
        if d["refresh"]: continue
        for key in ("thr_sub","thr_add"): rows += [int(r) for r in d[key]]
        for key in ("psq_sub","psq_add"): rows += [100000+int(r) for r in d[key]]
    recs.append((piece,fr,to,rows))
print("moves",len(recs), "rows/move", sum(len(r[3]) for r in recs)/len(recs))
def sim(order, cap, chunk=64, nx=8):
    caches=[OrderedDict() for _ in range(nx)]
    hits=tot=0
    for idx,ri in enumerate(order):
        x=(idx//chunk)%nx
        c=caches[x]
        for r in recs[ri][3]:
            tot+=1
            if r in c:
                hits+=1; c.move_to_end(r)
            else:
                c[r]=1
                if len(c)>cap: c.popitem(last=False)
    return hits/tot
n=len(recs)
orders={
 "as generated": list(range(n)),
 "by (piece, to)": sorted(range(n), key=lambda i:(recs[i][0],recs[i][2])),
 "by (piece, to, from)": sorted(range(n), key=lambda i:(recs[i][0],recs[i][2],recs[i][1])),
 "by (to, piece)": sorted(range(n), key=lambda i:(recs[i][2],recs[i][0])),
 "by (piece, from)": sorted(range(n), key=lambda i:(recs[i][0],recs[i][1])),
 "by (piece type, from)": sorted(range(n), key=lambda i:(recs[i][0]>>1,recs[i][1])),
 "by from": sorted(range(n), key=lambda i:(recs[i][1])),
}
for cap in (2048, 3072, 4096):
    for name,o in orders.items():
        print("cache %d rows/XCD  %-22s hit rate %.3f"%(cap,name,sim(o,cap)))
