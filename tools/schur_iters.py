"""Diagnostic: Francis steps per degree of the polynomials timeScaling's candidates c3..c8 solve for the lanes that reach
them (queue B) — the iteration count of the slowest lane is the latency of k_scaling_slow.
usage: python tools/schur_iters.py [n] [limits]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longtermplanner_amd import LongTermPlanner, limit_set
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
name = sys.argv[2] if len(sys.argv) > 2 else "panda"
dof, lim = limit_set(name)
ltp = LongTermPlanner(dof, 0.001, device=0, **lim)
qg, q0, v0, a0 = [x.cpu().numpy().reshape(n, dof) for x in ltp.generateQueries(n)]
O = oracle.Oracle(dof, 0.001, **lim)
r = O.plan_batch(qg, q0, v0, a0)
tot = {4: [], 5: [], 6: []}
items = 0
for p in range(n):
    if r["traj_len"][p] <= 0: continue
    for j in range(dof):
        if j == r["slowest"][p]: continue
        (ok, t, vd, m, case), polys = oracle.poly_log(lambda: O.time_scaling(j, qg[p, j], q0[p, j], v0[p, j], a0[p, j], r["dir"][p, j], r["t_required"][p]), cap=64)
        if case in (1, 2): continue
        items += 1
        for row in polys:
            d = int(row[0])
            oracle.roots_f64(row[1:2 + d])
            tot[d].append(oracle.lib().ltpo_last_schur_iterations())
print(f"{items} lanes beyond c1/c2")
for d, v in tot.items():
    v = np.array(v)
    if v.size: print(f"degree {d}: {v.size} solves, Francis steps mean {v.mean():.1f}, p50 {np.percentile(v,50):.0f}, p99 {np.percentile(v,99):.0f}, max {v.max()}; not converged: {(v > 40*d).sum()}")
