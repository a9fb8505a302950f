"""Parity report of the HIP path against the CPU oracle, in the form SURVEY.md §8(d) asks for:
max |dt| over the switching-time records, max |d(q,v,a,j)| over dense trajectories, the fraction of queries within
1e-9, and the outliers with a cause class. Writes one JSON object (default gpurun_out/parity_report.json).

  python tools/parity_report.py [n_records] [n_dense] [out.json]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import longtermplanner_amd as amd
import oracle

TOL = 1e-9
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_dense = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/parity_report.json"
PARTS = 16


def cause(dev, orc, i):
    """Why query i differs: the three classes SURVEY.md §8(d) lists, else 'rounding'."""
    if np.any(dev["mod"][i] != orc["mod"][i]) or np.any(dev["dir"][i] != orc["dir"][i]) or dev["slowest"][i] != orc["slowest"][i]:
        dv = np.abs(dev["v_drive"][i] - orc["v_drive"][i])
        return "root-classification" if np.any(np.isfinite(dv) & (dv > 1e-6)) else "window-test flip"
    if dev["traj_len"][i] != orc["traj_len"][i]:
        return "sample-index flip"
    return "rounding"


report = {"tolerance": TOL, "sets": {}}
for name in ("panda", "ref", "ref30"):
    D, lim = amd.limit_set(name)
    n = n_rec if D == 7 else n_rec // 10
    nd = n_dense if D == 7 else n_dense // 8
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    orc = oracle.Oracle(D, 0.001, **lim)
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=2026)
    t0 = time.time()
    dev = ltp.planBatchHost(qg, q0, v0, a0, sample=False)
    with ThreadPoolExecutor(PARTS) as ex:
        outs = list(ex.map(lambda i: orc.plan_batch(qg[i::PARTS], q0[i::PARTS], v0[i::PARTS], a0[i::PARTS], sample=False), range(PARTS)))
    ref = {k: np.empty_like(dev[k]) for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len")}
    ref["status"] = np.empty(n, dtype=np.int32)
    for i, o in enumerate(outs):
        for k in ref:
            ref[k][i::PARTS] = o[k]
    planned = ref["status"] != 0                      # oracle: 0 = rejected before sampling
    dev_planned = (dev["status"] & 7) == 0
    bad = planned != dev_planned
    worst = {}
    for k in ("t_opt", "t_scaled", "v_drive", "t_required"):
        d = np.abs(dev[k] - ref[k])
        same = (dev[k] == ref[k]) | (np.isnan(dev[k]) & np.isnan(ref[k]))      # inf == inf, NaN ~ NaN
        d = np.where(same, 0.0, d)
        d = np.where(np.isfinite(d), d, np.inf)
        d = d.reshape(n, -1).max(axis=1)
        d[~planned] = 0.0
        worst[k] = float(d.max())
        bad |= d > TOL
    for k in ("dir", "mod", "slowest", "traj_len"):
        neq = (dev[k] != ref[k]).reshape(n, -1).any(axis=1)
        bad |= neq & planned

    # dense q/v/a/j of the first nd plans
    sub = slice(0, nd)
    dd = ltp.planBatchHost(qg[sub], q0[sub], v0[sub], a0[sub], sample=True)
    dense = {"q": 0.0, "v": 0.0, "a": 0.0, "j": 0.0}
    samples = 0
    dense_bad = 0
    end_flag_mismatch = 0
    for p in range(nd):
        L = int(dd["traj_len"][p])
        if ref["status"][p] == 0:
            continue
        o = orc.plan_trajectory(qg[p], q0[p], v0[p], a0[p])
        if o["length"] != L:
            dense_bad += 1
            continue
        arrs = amd.unpack_trajectory(dd["packed"], int(dd["offsets"][p]), D, L)
        over = False
        for key, got in zip("qvaj", arrs):
            d = float(np.max(np.abs(got - o[key]))) if L else 0.0
            dense[key] = max(dense[key], d)
            over |= d > TOL
        dense_bad += int(over)
        samples += 4 * D * L
        end_flag_mismatch += int((o["status"] == 2) != bool(dd["status"][p] & 8))
    idx = np.nonzero(bad)[0]
    report["sets"][name] = {
        "dof": D, "t_sample": 0.001, "queries": int(n), "joint_lanes": int(n * D),
        "rejected_by_checkInputs_or_opt": int(np.sum(~planned)),
        "max_abs_dt": {k: worst[k] for k in worst},
        "integer_fields_equal": bool(not np.any([(dev[k] != ref[k]).reshape(n, -1).any(axis=1)[planned].any() for k in ("dir", "mod", "slowest", "traj_len")])),
        "fraction_within_tolerance": float(1.0 - idx.size / n),
        "outliers": [{"query": int(i), "cause": cause(dev, ref, i)} for i in idx[:50]],
        "dense": {"plans": int(nd), "values_compared": int(samples), "max_abs_d": dense,
                  "plans_beyond_tolerance_or_length_mismatch": int(dense_bad),
                  "end_limit_false": int(np.sum((dd["status"] & 8) != 0)), "end_limit_flag_mismatches": int(end_flag_mismatch)},
        "seconds": round(time.time() - t0, 1),
    }
    print(name, json.dumps(report["sets"][name]), flush=True)

os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(report, f, indent=1)
print("wrote", out_path)
