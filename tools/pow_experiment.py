"""Where do the last-bit differences between the device and the CPU oracle come from?

Claim under test (round-3 review, task 1): the ONLY arithmetic difference between the HIP path and the oracle is libm's
pow(x, 3 | 4 | 6) / pow(x, 0.5). The oracle has a test-only twin built with -DLTPO_EXACT_POW whose powers follow the device's
rule (one rounding of the exact product, csrc/ltp_math.hpp; sqrt for the exponent 1/2). If the claim holds,

  (A) the device's records (t_opt, t_scaled, v_drive, t_required, dir, mod, slowest, traj_len, status) are BIT-identical to the
      exact-pow twin's on every query, and
  (B) the default (libm) oracle differs from the twin exactly where it differs from the device.

(C) then bounds libm vs twin: max |dt| per branch ("site") of optSwitchTimes, found by re-running the joints that differ
through both libraries' one-joint entry points with the oracle's site diagnostics (oracle/ltp_oracle.c: ltpo_last_sites).

  python tools/pow_experiment.py [queries_per_7dof_set] [fuzz_sets] [queries_per_fuzz_set] [out.json] [seed ...]
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import longtermplanner_amd as amd
import oracle

argv = sys.argv[1:]
N = int(argv[0]) if len(argv) > 0 else 4_000_000
N_FUZZ_SETS = int(argv[1]) if len(argv) > 1 else 24
N_FUZZ = int(argv[2]) if len(argv) > 2 else 100_000
OUT = argv[3] if len(argv) > 3 else "gpurun_out/pow_experiment.json"
SEEDS = [int(x) for x in argv[4:]] or [9001]
THREADS = max(1, min(len(os.sched_getaffinity(0)), 32))
FLOAT_FIELDS = ("t_opt", "t_scaled", "v_drive", "t_required")
INT_FIELDS = ("dir", "mod", "slowest", "traj_len")
SITES = {1: "optBraking without phase 2 (cc:685-689)", 2: "modified profile (cc:119-124)", 4: "phase 2 absent: sqrt(root) (cc:130-142)",
         8: "phase 6 absent: sqrt(v_drive / j_max) (cc:150-162)", 16: "no cruise phase: 22-term radicand + pow(root, 1/2) (cc:192-243)",
         32: "quartic site A (cc:245-270)", 64: "acceleration limit after site A (cc:276-304)", 128: "quartic site B (cc:306-333)",
         256: "|q_diff| < eps early exit (cc:104-109)"}


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def same_bits(a, b):
    """elementwise: identical bit patterns, or NaN on both sides (the payload of a NaN is not part of the contract)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype == np.float64:
        return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
    return a == b


def oracle_batch(orc, q, parts=THREADS):
    n = q[0].shape[0]
    cuts = [n * i // parts for i in range(parts + 1)]
    with ThreadPoolExecutor(parts) as ex:
        outs = list(ex.map(lambda i: orc.plan_batch(*[x[cuts[i]:cuts[i + 1]] for x in q], sample=False), range(parts)))
    return {k: np.concatenate([o[k] for o in outs]) for k in FLOAT_FIELDS + INT_FIELDS + ("status",)}


def site_report(D, lim, Ts, q, dev, ol, oe, cap=12):
    """joints whose libm and exact-pow switching times differ most: which branch of optSwitchTimes they went through"""
    a = oracle.Oracle(D, Ts, **lim)
    b = oracle.Oracle(D, Ts, exact_pow=True, **lim)
    ok = ol["status"] != 0
    d = np.abs(ol["t_scaled"] - oe["t_scaled"])
    d[~np.isfinite(d)] = 0.0
    d[~ok] = 0.0
    dj = d.max(axis=2)                                    # [n][D]
    per_site = {}
    worst = []
    order = np.argsort(dj, axis=None)[::-1]
    nz = int(np.count_nonzero(dj))
    for flat in order[:min(nz, 4000)]:
        p, j = divmod(int(flat), D)
        qg, q0, v0, a0 = (float(x[p, j]) for x in q)
        slowest = int(ol["slowest"][p])
        sites = []
        for orc in (a, b):
            if j == slowest:
                orc.opt_switch_times(j, qg, q0, v0, a0, float(orc.v_max[j]))
            else:
                orc.time_scaling(j, qg, q0, v0, a0, float(ol["dir"][p, j]), float(ol["t_required"][p]))
            root = np.zeros(2)
            orc._lib.ltpo_last_site_root(root.ctypes.data_as(C.POINTER(C.c_double)))
            sites.append((int(orc._lib.ltpo_last_sites()), float(root[0]), float(root[1])))
        key = sites[0][0]
        e = per_site.setdefault(key, {"joints": 0, "max_abs_dt": 0.0})
        e["joints"] += 1
        e["max_abs_dt"] = max(e["max_abs_dt"], float(dj[p, j]))
        if len(worst) < cap:
            # how much ONE ulp of v_drive moves the switching times of this joint (time-scaled joints: timeScaling's closed-form
            # v_drive, cc:378-446, is where pow(x, 3 | 4) enters; optSwitchTimes then divides by v_drive and takes square roots)
            sens = None
            vd_l, vd_e = float(ol["v_drive"][p, j]), float(oe["v_drive"][p, j])
            if j != slowest and np.isfinite(vd_l) and vd_l > 0:
                dirj = float(ol["dir"][p, j])
                t1 = a.opt_switch_times(j, qg, q0, dirj * v0, dirj * a0, vd_l)[1]
                t2 = a.opt_switch_times(j, qg, q0, dirj * v0, dirj * a0, float(np.nextafter(vd_l, np.inf)))[1]
                sens = float(np.nanmax(np.abs(t2 - t1)))
            worst.append({"v_drive_bits_equal_libm_vs_exact": vd_l == vd_e, "v_drive_ulps_apart": (abs(vd_l - vd_e) / float(np.spacing(abs(vd_l)))) if vd_l != vd_e else 0.0,
                          "abs_dt_per_ulp_of_v_drive": sens,"query": p, "joint": j, "slowest_joint": j == slowest, "max_abs_dt_libm_vs_exact": float(dj[p, j]),
                          "max_abs_dt_device_vs_libm": float(np.nanmax(np.abs(dev["t_scaled"][p, j] - ol["t_scaled"][p, j]))),
                          "device_bits_equal_exact": bool(np.all(same_bits(dev["t_scaled"][p, j], oe["t_scaled"][p, j]))),
                          "sites_libm": sites[0][0], "sites_exact": sites[1][0],
                          "site16_radicand": sites[0][1], "site16_radicand_scale": sites[0][2],
                          "site16_amplification": (sites[0][2] / sites[0][1]) if (sites[0][0] & 16) and sites[0][1] > 0 else None,
                          "t_required": float(ol["t_required"][p]), "v_drive": float(ol["v_drive"][p, j])})
    named = []
    for key, e in sorted(per_site.items(), key=lambda kv: -kv[1]["max_abs_dt"]):
        named.append({"sites": key, "branches": [v for k, v in SITES.items() if key & k] or ["standard profile, all phases"], **e})
    return {"joints_that_differ": nz, "joints_classified": int(min(nz, 4000)), "by_branch": named, "worst": worst}


def run_set(name, D, lim, Ts, n, seed):
    t0 = time.time()
    ltp = amd.LongTermPlanner(D, Ts, device=0, **lim)
    ltp.setPowRule("exact")                       # this experiment is about the correctly rounded powers (the opt-in rule)
    q = amd.generate_queries(n, lim, seed=seed)
    dev = ltp.planBatchHost(*q, sample=False)
    t1 = time.time()
    ol = oracle_batch(oracle.Oracle(D, Ts, **lim), q)
    oe = oracle_batch(oracle.Oracle(D, Ts, exact_pow=True, **lim), q)
    t2 = time.time()
    res = {"dof": D, "t_sample": Ts, "seed": seed, "queries": n, "joint_lanes": n * D}
    dev_ok = (dev["status"] & 0x57) == 0                          # planned (END_LIMIT is not part of the records' verdict here)
    res["verdict_equal_exact"] = bool(np.array_equal(dev_ok, oe["status"] != 0))
    res["verdict_equal_libm"] = bool(np.array_equal(dev_ok, ol["status"] != 0))
    ok = dev_ok & (oe["status"] != 0) & (ol["status"] != 0)
    cmp_exact, cmp_libm, l_vs_e = {}, {}, {}
    all_identical = res["verdict_equal_exact"]
    for k in FLOAT_FIELDS + INT_FIELDS:
        dk = dev[k][ok].astype(oe[k].dtype) if k in INT_FIELDS else dev[k][ok]
        ne = ~same_bits(dk, oe[k][ok])
        cmp_exact[k] = {"entries": int(ne.size), "not_bit_identical": int(ne.sum())}
        all_identical = all_identical and not ne.any()
        if k in FLOAT_FIELDS:
            for tag, other, into in (("dev", ol, cmp_libm), ("lvse", None, l_vs_e)):
                x, y = (dev[k][ok], ol[k][ok]) if tag == "dev" else (ol[k][ok], oe[k][ok])
                d = np.abs(x - y)
                d = d[np.isfinite(d)]
                into[k] = {"max_abs_d": float(d.max()) if d.size else 0.0, "entries_that_differ": int((~same_bits(x, y)).sum()),
                           "beyond_1e-9": int((d > 1e-9).sum())}
            # (B): libm differs from the twin exactly where it differs from the device
            cmp_libm[k]["same_entries_as_libm_vs_exact"] = bool(np.array_equal(~same_bits(dev[k][ok], ol[k][ok]), ~same_bits(ol[k][ok], oe[k][ok])))
        else:
            cmp_libm[k] = {"not_equal": int((dk != ol[k][ok]).sum())}
            l_vs_e[k] = {"not_equal": int((ol[k][ok] != oe[k][ok]).sum())}
    res["device_vs_exact_pow_oracle"] = {"all_records_bit_identical": bool(all_identical), "fields": cmp_exact}
    res["device_vs_libm_oracle"] = cmp_libm
    res["libm_oracle_vs_exact_pow_oracle"] = l_vs_e
    res["amplification"] = site_report(D, lim, Ts, q, dev, ol, oe)
    res["seconds"] = {"device_and_copies": round(t1 - t0, 1), "two_oracle_passes": round(t2 - t1, 1), "total": round(time.time() - t0, 1)}
    print(name, json.dumps({k: v for k, v in res.items() if k != "amplification"}), flush=True)
    print(name, "amplification", json.dumps(res["amplification"]["by_branch"][:6]), flush=True)
    return res


def main():
    report = {"what": "device records vs the exact-pow twin of the oracle (bit for bit), vs the libm oracle, and libm vs twin by branch",
              "host_threads": THREADS, "sets": {}}
    for seed in SEEDS:
        for name in ("panda", "ref", "ref30"):
            D, lim = amd.limit_set(name)
            n = N if D == 7 else N // 5
            if n > 0:
                report["sets"][f"{name} seed {seed}"] = run_set(name, D, lim, 0.001, n, seed)
    rng = np.random.default_rng(4041)
    for trial in range(N_FUZZ_SETS):
        # wider than round 3's fuzz: j_max / Ts up to 1e9, Ts down to 0.1 ms, slow-jerk sets with long trajectories
        D = int(rng.integers(1, 13))
        ts = float(rng.choice([0.0001, 0.00025, 0.001, 0.002, 0.004, 0.01]))
        v_max = rng.uniform(0.5, 3.0, D)
        a_max = rng.uniform(1.0, 20.0, D)
        kind = trial % 3
        if kind == 0:
            j_max = a_max * rng.uniform(5.0, 600.0, D)
        elif kind == 1:
            j_max = np.minimum(a_max * rng.uniform(500.0, 5000.0, D), 1e9 * ts)      # j_max / Ts up to 1e9
        else:
            j_max = a_max * rng.uniform(0.05, 2.0, D)                                 # slow jerk: trajectories of 1e4-1e5 samples
        q_hi = rng.uniform(1.0, 3.5, D)
        lim = dict(q_min=list(-q_hi), q_max=list(q_hi), v_max=list(v_max), a_max=list(a_max), j_max=list(j_max))
        r = run_set(f"fuzz{trial}", D, lim, ts, N_FUZZ, 7000 + trial)
        r["limits"] = {k: [float(x) for x in v] for k, v in lim.items()}
        r["j_max_over_ts_max"] = float(np.max(j_max) / ts)
        report["sets"][f"fuzz{trial}"] = r
    sets = report["sets"].values()
    report["total_queries"] = int(sum(s["queries"] for s in sets))
    report["total_joint_lanes"] = int(sum(s["joint_lanes"] for s in sets))
    report["all_sets_device_bit_identical_to_exact_pow_oracle"] = bool(all(s["device_vs_exact_pow_oracle"]["all_records_bit_identical"] for s in sets))
    report["worst_libm_vs_exact_abs_dt"] = float(max(s["libm_oracle_vs_exact_pow_oracle"]["t_scaled"]["max_abs_d"] for s in sets))
    os.makedirs(os.path.dirname(os.path.abspath(OUT)), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", OUT, "-", report["total_queries"], "queries; device == exact-pow oracle bit for bit:",
          report["all_sets_device_bit_identical_to_exact_pow_oracle"], "; worst libm vs exact |dt|", report["worst_libm_vs_exact_abs_dt"])


if __name__ == "__main__":
    main()
