"""Condenses the round's parity runs into profiles/rNN_parity_report.json:
  tools/pow_experiment.py          -> records: device vs the exact-pow twin (bit for bit), vs the libm oracle, libm vs twin by branch
  tools/dense_soak.py [--wide-fuzz] -> dense: every q/v/a/j sample vs the libm oracle (the parity reference)
  tools/dense_soak.py --exact-pow   -> dense: the same vs the twin (diagnostic)
usage: python tools/collect_parity_report.py r04 gpurun_out/r04_pow_experiment.json gpurun_out/r04_dense_libm.json gpurun_out/r04_dense_exact.json
(aggregates profiles/rNN_dense_soak_*_{more_seeds,wide_fuzz,matlab}*.json from tools/aggregate_dense_soaks.py are added as totals when present)"""
import json
import sys

tag, f_pow, f_libm, f_exact = sys.argv[1:5]
pw, dl, de = (json.load(open(f)) for f in (f_pow, f_libm, f_exact))


def dense_summary(r):
    out = {"oracle": r["oracle"], "dense_trajectories": r["total_dense_plans"], "values_compared": r["total_values_compared"], "tolerance": r["tolerance"],
           "host_threads": r["host_threads"], "sets": {}}
    for k, s in r["sets"].items():
        out["sets"][k] = {x: s[x] for x in ("dense_plans", "sampled", "values_compared", "max_abs_d", "plans_beyond_tolerance", "verdict_mismatches",
                                            "length_mismatches", "end_limit_flag_mismatches", "plans_with_bit_identical_jerk_rows",
                                            "outliers_examined", "outliers_explained_by_dt", "outliers") if x in s}
        if "sets" in s:
            out["sets"][k]["limit_sets"] = s["sets"]
    tot = lambda key: sum(s.get(key, 0) for s in r["sets"].values())
    out["totals"] = {"sampled": tot("sampled"), "plans_beyond_tolerance": tot("plans_beyond_tolerance"), "outliers_explained_by_dt": tot("outliers_explained_by_dt"),
                     "verdict_mismatches": tot("verdict_mismatches"), "length_mismatches": tot("length_mismatches"),
                     "end_limit_flag_mismatches": tot("end_limit_flag_mismatches"), "plans_with_bit_identical_jerk_rows": tot("plans_with_bit_identical_jerk_rows"),
                     "max_abs_d": {c: max(s["max_abs_d"][c] for s in r["sets"].values()) for c in "qvaj"}}
    return out


records = {"what": pw["what"], "host_threads": pw["host_threads"], "queries": pw["total_queries"], "joint_lanes": pw["total_joint_lanes"],
           "device_bit_identical_to_exact_pow_oracle_in_every_set": pw["all_sets_device_bit_identical_to_exact_pow_oracle"],
           "worst_abs_dt_libm_vs_exact_pow": pw["worst_libm_vs_exact_abs_dt"], "sets": {}}
by_branch, worst = {}, []
for k, s in pw["sets"].items():
    records["sets"][k] = {"dof": s["dof"], "t_sample": s["t_sample"], "queries": s["queries"], "j_max_over_ts_max": s.get("j_max_over_ts_max"),
                          "verdicts_equal": s["verdict_equal_exact"] and s["verdict_equal_libm"],
                          "device_vs_exact_pow_oracle": {"all_records_bit_identical": s["device_vs_exact_pow_oracle"]["all_records_bit_identical"],
                                                         "entries_not_bit_identical": sum(v["not_bit_identical"] for v in s["device_vs_exact_pow_oracle"]["fields"].values())},
                          "device_vs_libm_oracle": s["device_vs_libm_oracle"], "libm_oracle_vs_exact_pow_oracle": s["libm_oracle_vs_exact_pow_oracle"]}
    for b in s["amplification"]["by_branch"]:
        e = by_branch.setdefault(str(b["sites"]), {"branches": b["branches"], "joints": 0, "max_abs_dt": 0.0})
        e["joints"] += b["joints"]
        e["max_abs_dt"] = max(e["max_abs_dt"], b["max_abs_dt"])
    worst += [dict(w, set=k) for w in s["amplification"]["worst"]]
worst.sort(key=lambda w: -w["max_abs_dt_libm_vs_exact"])
records["libm_vs_exact_pow_by_branch_of_optSwitchTimes"] = sorted(by_branch.values(), key=lambda e: -e["max_abs_dt"])
records["worst_joints_libm_vs_exact_pow"] = worst[:16]
records["where_one_ulp_becomes_1e-13_s"] = (
    "Not inside optSwitchTimes: one ulp of v_drive moves a switching time by 1e-16 .. 3e-14 s (abs_dt_per_ulp_of_v_drive). The amplifier is "
    "timeScaling's closed-form v_drive (cc:378-396 standard profile, cc:408-446 modified profile): v_drive = (A - sqrt(B) / 12) / j_max with A and "
    "sqrt(B) / 12 nearly equal sums of ~20 terms of size a_max j_max t_required, B holding pow(a_0, 3), pow(a_0, 4), pow(a_max, 3), pow(a_max, 4): a last-bit "
    "difference of one pow() is a last-bit difference of B, and the cancellation A - sqrt(B) / 12 leaves it as 1e3 .. 1e5 ulps of a SMALL v_drive "
    "(v_drive_ulps_apart; worst 99 535 ulps at v_drive 0.30). 1e5 ulps x 9e-16 s per ulp = 4.5e-11 s, the worst |dt| seen.")
report = {"round": tag,
          "claims": {"A": "device records are BIT-identical to the oracle built with the device's pow rule (-DLTPO_EXACT_POW) on every query",
                     "B": "the libm oracle differs from that twin exactly where it differs from the device (same entries)",
                     "C": "hence libm's pow(x, 3 | 4 | 6) / pow(x, 0.5) is the ONLY source of device-vs-reference differences in the records; the jerk "
                          "rows are bit-identical to the twin's in every plan, and against the libm oracle every a / j sample beyond 1e-9 is a "
                          "jerk-correction sample (cc:768-807) that carries |dt| x j_max / Ts of a plan whose |dt| <= 1e-9"},
          "records": records, "dense_vs_libm_oracle": dense_summary(dl), "dense_vs_exact_pow_oracle": dense_summary(de)}
# further dense soaks of the round, condensed by tools/aggregate_dense_soaks.py (other seeds, wide fuzz, MATLAB semantics): totals only
import os
more = {}
for key, stem in (("more_seeds", "dense_soak_{}_more_seeds"), ("wide_fuzz", "dense_soak_wide_fuzz_{}"), ("matlab_semantics", "dense_soak_matlab_{}")):
    for oracle_key, oracle_name in (("vs_libm_oracle", "libm"), ("vs_exact_pow_oracle", "exact_pow")):
        path = f"profiles/{tag}_" + stem.format(oracle_name) + ".json"
        if os.path.exists(path):
            a = json.load(open(path))
            more.setdefault(key, {})[oracle_key] = {"file": path, "runs": len(a["runs"]), "total": a["total"], "max_abs_d": a["max_abs_d"],
                                                    "fraction_of_plans_within_tolerance": a["fraction_of_plans_within_tolerance"]}
if more:
    report["dense_further_soaks"] = more
out = f"profiles/{tag}_parity_report.json"
with open(out, "w") as f:
    json.dump(report, f, indent=1)
print("wrote", out)
