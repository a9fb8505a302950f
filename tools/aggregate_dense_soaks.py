"""Sums the reports of several tools/dense_soak.py runs (other seeds) into one JSON: totals, worst differences, every outlier.

  python tools/aggregate_dense_soaks.py out.json report1.json report2.json ...
"""
import json
import sys


def main():
    out_path, files = sys.argv[1], sys.argv[2:]
    keys = ("dense_plans", "sampled", "values_compared", "plans_beyond_tolerance", "verdict_mismatches", "length_mismatches",
            "end_limit_flag_mismatches", "plans_with_bit_identical_jerk_rows", "outliers_examined", "outliers_explained_by_dt")
    total = {k: 0 for k in keys}
    worst = {k: 0.0 for k in "qvaj"}
    runs, outliers, oracle, tol = [], [], None, None
    for f in files:
        d = json.load(open(f))
        oracle, tol = d["oracle"], d["tolerance"]
        run = {"report": f.split("/")[-1], "dense_plans": d["total_dense_plans"], "values_compared": d["total_values_compared"], "plans_beyond_tolerance": 0}
        for name, v in d["sets"].items():
            for k in keys:
                total[k] += v[k]
            run["plans_beyond_tolerance"] += v["plans_beyond_tolerance"]
            for k in "qvaj":
                worst[k] = max(worst[k], v["max_abs_d"][k])
            for o in v["outliers"]:
                outliers.append({"report": run["report"], "set": name.split(" ")[0] if "set" not in o else f"fuzz{o['set']}", "query": o["query"],
                                 "max_abs_d": o["max_abs_d"], "max_abs_dt": o.get("max_abs_dt"), "cause": o["cause"],
                                 "explained_by_dt_times_jmax_over_ts": o.get("explained_by_dt_times_jmax_over_ts")})
        runs.append(run)
    rep = {"what": "tools/dense_soak.py over other query seeds and other fuzzed limit sets than the committed report's: every q/v/a/j sample of the device's dense rows vs the oracle's planTrajectory",
           "oracle": oracle, "tolerance": tol, "runs": runs, "total": total, "max_abs_d": worst,
           "fraction_of_plans_within_tolerance": 1.0 - total["plans_beyond_tolerance"] / max(total["sampled"], 1),
           "outliers": outliers}
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: rep[k] for k in ("oracle", "total", "max_abs_d", "fraction_of_plans_within_tolerance")}))


if __name__ == "__main__":
    main()
