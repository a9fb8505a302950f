"""Dense-trajectory parity soak: EVERY sample of q/v/a/j of the HIP path against the CPU oracle, on ~10^6 plans.

For each limit set the batch is planned and sampled on the device chunk by chunk (a reused tile, as bench.py does), each
chunk's rows are copied to pinned host memory, and 16 host threads run the oracle's full planTrajectory (cc:7-63) on the
same queries and compare sample by sample (oracle/ltp_oracle.c: ltpo_compare_dense). Plans beyond the tolerance are listed
with a cause class, as SURVEY.md §8(d) asks: root-classification, window-test flip, sample-index flip, else rounding.

  python tools/dense_soak.py [--exact-pow] [--wide-fuzz] [--matlab] [--pow-rule-libm] [panda_plans] [ref_plans] [ref30_plans] [fuzz_sets] [plans_per_fuzz_set] [out.json] [seed_shift]

--exact-pow: compare against the oracle's DIAGNOSTIC twin (-DLTPO_EXACT_POW: the device's rule for pow(x, 3 | 4 | 6) and
pow(x, 0.5) restated in C) instead of the libm oracle. If libm's pow is the only source of last-bit differences, the jerk rows
are then bit-identical in every plan and nothing lies beyond the tolerance (tools/pow_experiment.py has the records' side).
--pow-rule-libm: the device under the pow rule LTP_POW_LIBM (glibc's pow restated): against the libm oracle nothing may lie beyond
the tolerance and every jerk row must be bit-identical.
--matlab: device and oracle in MATLAB semantics (LTPlanner.m); the device's rows come from k_sample_walk_matlab_*.
--wide-fuzz: limit sets with j_max / Ts up to 1e9, Ts down to 0.1 ms and slow-jerk sets whose trajectories have 1e4-1e5 samples.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import longtermplanner_amd as amd
from dense_compare import THREADS, TOL, fuzz_limits, pinned_buffers, soak as _soak

EXACT = "--exact-pow" in sys.argv[1:]
WIDE = "--wide-fuzz" in sys.argv[1:]
MATLAB = "--matlab" in sys.argv[1:]
LIBM_RULE = "--pow-rule-libm" in sys.argv[1:]      # the device forms its powers as glibc's pow does (ltp_set_pow_rule(LTP_POW_LIBM))
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
n_panda = int(argv[0]) if len(argv) > 0 else 600_000
n_ref = int(argv[1]) if len(argv) > 1 else 300_000
n_ref30 = int(argv[2]) if len(argv) > 2 else 50_000
n_fuzz_sets = int(argv[3]) if len(argv) > 3 else 24
n_fuzz = int(argv[4]) if len(argv) > 4 else 2_500
out_path = argv[5] if len(argv) > 5 else "gpurun_out/dense_soak.json"
seed_shift = int(argv[6]) if len(argv) > 6 else 0          # other query sets than the committed report's (seeds 4242 / 5000 + trial)


def soak(name, D, lim, Ts, n, seed, bufs):
    return _soak(name, D, lim, Ts, n, seed, bufs, exact=EXACT, matlab=MATLAB, pow_rule="libm" if LIBM_RULE else "exact")




def main():
    bufs = pinned_buffers()
    report = {"tolerance": TOL, "host_threads": THREADS, "semantics": "matlab" if MATLAB else "cpp", "device_pow_rule": "libm" if LIBM_RULE else "exact",
              "oracle": "exact-pow twin (diagnostic)" if EXACT else "libm (the parity reference)",
              "what": "every q/v/a/j sample of the device's dense rows vs the oracle's planTrajectory", "sets": {}}
    for name, n in (("panda", n_panda), ("ref", n_ref), ("ref30", n_ref30)):
        if n > 0:
            D, lim = amd.limit_set(name)
            report["sets"][name] = soak(name, D, lim, 0.001, n, 4242 + seed_shift, bufs)
    rng = np.random.default_rng(2027 + seed_shift)
    fuzz = {"sets": [], "dense_plans": 0, "sampled": 0, "outliers_examined": 0, "outliers_explained_by_dt": 0, "values_compared": 0, "plans_beyond_tolerance": 0, "max_abs_d": {k: 0.0 for k in "qvaj"},
            "verdict_mismatches": 0, "length_mismatches": 0, "end_limit_flag_mismatches": 0, "plans_with_bit_identical_jerk_rows": 0, "outliers": []}
    for trial in range(n_fuzz_sets):
        D, ts, lim = fuzz_limits(rng, trial, WIDE)
        j_max = np.asarray(lim["j_max"])
        r = soak(f"fuzz{trial}", D, lim, ts, n_fuzz, 5000 + trial + 100 * seed_shift, bufs)
        fuzz["sets"].append({"dof": D, "t_sample": ts, "j_max_over_ts_max": float(np.max(j_max) / ts), "max_abs_d": r["max_abs_d"],
                             "plans_beyond_tolerance": r["plans_beyond_tolerance"], "sampled": r["sampled"]})
        for k in ("dense_plans", "sampled", "values_compared", "plans_beyond_tolerance", "verdict_mismatches", "length_mismatches", "end_limit_flag_mismatches",
                  "plans_with_bit_identical_jerk_rows", "outliers_examined", "outliers_explained_by_dt"):
            fuzz[k] += r[k]
        for k in "qvaj":
            fuzz["max_abs_d"][k] = max(fuzz["max_abs_d"][k], r["max_abs_d"][k])
        fuzz["outliers"] += [dict(o, set=trial) for o in r["outliers"]]
    if n_fuzz_sets:
        report["sets"]["fuzzed limit sets (dof 1-12, Ts 1-10 ms, random per-joint limits)"] = fuzz
    report["total_dense_plans"] = sum(v["dense_plans"] for v in report["sets"].values())
    report["total_values_compared"] = sum(v["values_compared"] for v in report["sets"].values())
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", out_path, "-", report["total_dense_plans"], "dense plans,", report["total_values_compared"], "values")


if __name__ == "__main__":
    main()
