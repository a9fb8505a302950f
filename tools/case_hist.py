"""Diagnostic: which timeScaling candidate (cc:378-638) wins, per (query, joint), for the bench's synthetic queries — from the
oracle, on the host. usage: python tools/case_hist.py [n] [limits]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longtermplanner_amd import LongTermPlanner, limit_set
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else "panda"
dof, lim = limit_set(name)
ltp = LongTermPlanner(dof, 0.001, device=0, **lim)
qg, q0, v0, a0 = [x.cpu().numpy().reshape(n, dof) for x in ltp.generateQueries(n)]
O = oracle.Oracle(dof, 0.001, **lim)
r = O.plan_batch(qg, q0, v0, a0)
hist = np.zeros(9, dtype=np.int64)
lanes = 0
for p in range(n):
    if r["status"][p] != 1 and r["n_ok"] >= 0 and r["traj_len"][p] <= 0: continue
    for j in range(dof):
        if j == r["slowest"][p]: continue
        ok, t, vd, m, case = O.time_scaling(j, qg[p, j], q0[p, j], v0[p, j], a0[p, j], r["dir"][p, j], r["t_required"][p])
        hist[case] += 1
        lanes += 1
print(f"{n} queries, {lanes} scaled joints; winning candidate (0 = none): {hist.tolist()}")
print("share needing more than c1/c2 (queue B):", float(hist[0] + hist[3:].sum()) / max(lanes, 1))
