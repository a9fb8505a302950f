"""Isolates the device eigen-solve (k_roots_probe, one wave = 64 polynomials, as one candidate wave of k_scaling_slow): the degree-4 /
5 / 6 polynomials the timeScaling candidates of a 100 k-query batch really solve, timed per degree with HIP events.
usage: python tools/roots_probe_bench.py [n] [limits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import longtermplanner_amd as amd
import oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
name = sys.argv[2] if len(sys.argv) > 2 else "panda"
dof, lim = amd.limit_set(name)
qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=12345)
O = oracle.Oracle(dof, 0.001, **lim)
_, polys = oracle.poly_log(lambda: O.plan_batch(qg, q0, v0, a0, want_records=False), cap=2000000)
ltp = amd.LongTermPlanner(dof, 0.001, device=0, **lim)
for deg in (4, 5, 6):
    rows = np.array([r[1:8] for r in polys if int(r[0]) == deg])
    if not len(rows):
        continue
    steps = []
    for r in rows:
        oracle.roots_f64(r[:deg + 1]); steps.append(oracle.lib().ltpo_last_schur_iterations())
    steps = np.array(steps)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = ltp.debugRootsProbe(deg, rows)
        best = min(best, time.perf_counter() - t0)
    want = np.array([oracle.smallest_root(r[:deg + 1]) for r in rows])
    same = np.array_equal(got, want) or np.all((got == want) | (np.isinf(got) & np.isinf(want)))
    print(f"degree {deg}: {len(rows)} polynomials, Francis steps mean {steps.mean():.1f} max {steps.max()}, host-side call {best*1e6:.0f} us (incl. copies), bit-identical to the oracle: {same}")
