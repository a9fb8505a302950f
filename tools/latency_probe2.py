import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import longtermplanner_amd as amd
D, lim = amd.limit_set("panda")
ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
qg, q0, v0, a0 = amd.generate_queries(64, lim, seed=3)
for name, fn in (("optBraking", lambda i: ltp.optBraking(0, v0[i, 0], a0[i, 0])),
                 ("optSwitchTimes", lambda i: ltp.optSwitchTimes(0, qg[i, 0], q0[i, 0], v0[i, 0], a0[i, 0], lim["v_max"][0])),
                 ("timeScaling", lambda i: ltp.timeScaling(0, qg[i, 0], q0[i, 0], v0[i, 0], a0[i, 0], 1.0, 2.0)),
                 ("checkInputs", lambda i: ltp.checkInputs(q0[i], v0[i], a0[i]))):
    for i in range(5): fn(i)
    ts = []
    for i in range(64):
        t0 = time.perf_counter(); fn(i); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{name:20s} mean {ts.mean():8.1f} median {np.median(ts):8.1f} min {ts.min():8.1f} max {ts.max():8.1f} us")
