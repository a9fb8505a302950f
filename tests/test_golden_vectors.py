"""Committed regression vectors (tests/golden/oracle_vectors.npz, made by tests/golden/make_oracle_vectors.py).

CPU: the oracle still reproduces them bit for bit (guards the checker itself against drift between rounds).
GPU: the HIP path matches them to 1e-9 through the C ABI — the comparison needs no oracle code at run time.
"""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "oracle_vectors.npz")


def _maker():
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(HERE, "golden", "make_oracle_vectors.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_committed_vectors(oracle_mod):
    want = np.load(VEC)
    got = _maker().build()
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        assert np.array_equal(want[k], got[k], equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["panda", "ref", "ref30"])
def test_device_matches_committed_vectors(name):
    import longtermplanner_amd as amd
    g = np.load(VEC)
    D, lim = amd.limit_set(name)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ins = [g[f"{name}/{k}"] for k in ("q_goal", "q_0", "v_0", "a_0")]
    r = ltp.planBatchHost(*ins, sample=True)
    assert np.array_equal(r["traj_len"], g[f"{name}/traj_len"])
    assert np.array_equal(r["status"] == 0, g[f"{name}/status"] == 1)
    assert np.array_equal(r["slowest"], g[f"{name}/slowest"]) and np.array_equal(r["mod"], g[f"{name}/mod"])
    for k in ("t_opt", "t_scaled", "v_drive", "t_required", "dir"):
        assert np.nanmax(np.abs(r[k] - g[f"{name}/{k}"])) <= 1e-9, k
    maker = _maker()
    for p in range(ins[0].shape[0]):
        L = int(r["traj_len"][p])
        if L <= 0:
            continue
        idx = maker.sample_indices(L)
        rows = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), D, L)
        for k, arr in enumerate(rows):
            assert np.max(np.abs(arr[:, idx] - g[f"{name}/samples"][k, p])) <= 1e-9, (p, k)


POLY = os.path.join(HERE, "golden", "planner_polynomials.npz")


def test_oracle_root_finder_reproduces_planner_polynomials(oracle_mod):
    """the polynomials the planner really solves (all degrees, incl. 'no admissible root')."""
    rows = np.load(POLY)["rows"]
    assert {4, 5, 6} <= set(rows[:, 0].astype(int)) and np.isinf(rows[:, 8]).sum() > 100
    for r in rows:
        d = int(r[0])
        got = oracle_mod.smallest_root(r[1:d + 2])
        assert got == r[8] or (np.isnan(got) and np.isnan(r[8])), r


def test_selected_roots_agree_with_lapack():
    """The restated Eigen 3.4 eigen-solve against an independent one (numpy.roots = LAPACK on the same companion
    matrix): same 'smallest positive exactly-real root > 1e-7' (roots.h:43-50) wherever the choice is well conditioned.
    LAPACK does not return exactly-real eigenvalues the way Eigen's real Schur form does, so its imaginary parts are
    compared against a tolerance and polynomials with a nearly-double real root (classification genuinely ambiguous)
    are counted, not compared."""
    rows = np.load(POLY)["rows"]
    ambiguous = compared = 0
    worst = 0.0
    for r in rows:
        d = int(r[0])
        p = r[1:d + 2]
        if not np.all(np.isfinite(p)) or p[0] == 0.0:
            continue
        z = np.roots(p)
        scale = np.maximum(1.0, np.abs(z))
        real = np.abs(z.imag) <= 1e-9 * scale
        near = (np.abs(z.imag) <= 1e-4 * scale) & ~real            # nearly-double real root / nearly-real pair
        cand = z.real[real & (z.real > 1e-7)]
        want = cand.min() if cand.size else np.inf
        if near.any() and (z.real[near] > 1e-7).any() and z.real[near & (z.real > 1e-7)].min() < want * (1 + 1e-6):
            ambiguous += 1
            continue
        compared += 1
        if np.isinf(want) or np.isinf(r[8]):
            assert np.isinf(want) and np.isinf(r[8]), (r, z)
        else:
            worst = max(worst, abs(r[8] - want) / max(1.0, abs(want)))
    assert compared > 0.95 * len(rows), (compared, ambiguous)
    assert worst < 1e-8, worst


@pytest.mark.gpu
def test_device_root_finder_on_planner_polynomials():
    import longtermplanner_amd as amd
    rows = np.load(POLY)["rows"]
    D, lim = amd.limit_set("panda")
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    for d in (4, 5, 6):
        sel = rows[rows[:, 0] == d]
        got = ltp.debugRootsProbe(d, sel[:, 1:8])
        assert np.array_equal(np.isinf(got), np.isinf(sel[:, 8])), d
        fin = np.isfinite(sel[:, 8])
        rel = np.abs(got[fin] - sel[fin, 8]) / np.maximum(1.0, np.abs(sel[fin, 8]))
        assert rel.max() <= 1e-12, (d, rel.max())
