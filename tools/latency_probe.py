"""Latency of the single-call (host pointer) entry points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import longtermplanner_amd as amd
D, lim = amd.limit_set("panda")
ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
qg, q0, v0, a0 = amd.generate_queries(64, lim, seed=3)
traj = amd.Trajectory()
for name, fn in (("planTrajectory (7-DoF, ~1700 samples)", lambda i: ltp.planTrajectory(qg[i], q0[i], v0[i], a0[i], traj)),
                 ("planBatchHost n=1 switch-only", lambda i: ltp.planBatchHost(qg[i], q0[i], v0[i], a0[i], sample=False)),
                 ("checkInputs", lambda i: ltp.checkInputs(q0[i], v0[i], a0[i])),
                 ("optSwitchTimes", lambda i: ltp.optSwitchTimes(0, qg[i, 0], q0[i, 0], v0[i, 0], a0[i, 0], lim["v_max"][0]))):
    for i in range(5): fn(i)
    t0 = time.perf_counter()
    for i in range(64): fn(i)
    print(f"{name:45s} {(time.perf_counter() - t0) / 64 * 1e6:9.1f} us per call")
