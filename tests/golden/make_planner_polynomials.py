#!/usr/bin/env python3
"""Regenerates tests/golden/planner_polynomials.npz: polynomials the planner really solves, with the root it selects.

SURVEY.md §8(c) item 3. Produced by THIS repository's CPU oracle (the reference cannot be built here): every
polynomial that planning random batches handed to roots() (reference call sites src/long_term_planner.cc:256-261,
316-321, 467-472, 508-513, 535-540, 561-566, 587-592, 622-627), thinned to a small fixture that keeps every degree-5
and degree-6 case and the "no admissible root -> +inf" cases. Rows: [degree, p0..p6 (highest power first), root].
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from longtermplanner_amd import generate_queries, limit_set  # noqa: E402


def build():
    logs = []
    for name, n in (("ref", 6000), ("panda", 6000), ("ref30", 600)):
        D, lim = limit_set(name)
        orc = oracle.Oracle(D, 0.001, **lim)
        qg, q0, v0, a0 = generate_queries(n, lim, seed=99)
        _, log = oracle.poly_log(lambda: orc.plan_batch(qg, q0, v0, a0, sample=False))
        logs.append(log)
    log = np.concatenate(logs)
    deg = log[:, 0].astype(int)
    keep = np.zeros(len(log), dtype=bool)
    keep[deg >= 5] = True
    quartic = np.nonzero(deg == 4)[0]
    no_root = quartic[np.isinf(log[quartic, 8])]
    keep[no_root[:300]] = True
    keep[quartic[:: max(1, len(quartic) // 1200)]] = True
    return log[keep]


if __name__ == "__main__":
    rows = build()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "planner_polynomials.npz"), rows=rows)
    d = rows[:, 0].astype(int)
    print("wrote planner_polynomials.npz:", {k: int((d == k).sum()) for k in (4, 5, 6)}, "no admissible root:", int(np.isinf(rows[:, 8]).sum()))
