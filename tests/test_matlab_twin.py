"""The MATLAB-semantics twin of the oracle: oracle.Oracle(..., semantics="matlab") follows LTPlanner.m
where it diverges from src/long_term_planner.cc. Pins, all reference-held or reproducible here:
  * the three MATLAB unit tables (tests/unittests/*.m), all seven switching times;
  * the two MATLAB grid tests (tests/gridTestOneJoint.m, gridTestTimeScaling.m): no scenario unfinished, failed or late, and
    the accuracy README.md:128-136 publishes from exactly that script (mean goal error 0.003 rad, worst below 0.015);
  * MATLAB's roots() output order, which LTPlanner.m indexes by position: restated LAPACK DGEEV path vs numpy.roots (the same
    LAPACK driver on the same companion matrix);
  * the sampler against an independent numpy restatement (cumsum form), including the last-joint-only acceleration tail.
Against MATLAB itself the mode is unpinned beyond these (no MATLAB / Octave in the image)."""
import ctypes as C

import numpy as np
import pytest


def _matlab(oracle_mod, dof, ts, v, a, j):
    z = [0.0] * dof
    return oracle_mod.Oracle(dof, ts, z, z, v, a, j, semantics="matlab")


def test_matlab_roots_order_is_lapacks(oracle_mod):
    rng = np.random.default_rng(11)
    rows = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "planner_polynomials.npz"))["rows"]
    for r in rows:                                           # polynomials the planner really solves: every one in numpy's order
        deg = int(r[0])
        m, st = oracle_mod.matlab_roots(r[1:2 + deg])
        w = np.roots(r[1:2 + deg])
        assert st == 0 and m.size == w.size
        assert np.array_equal(m.imag == 0, w.imag == 0)
        assert np.max(np.abs(m - w) / np.abs(w)) < 1e-9
    swapped = total = 0
    for deg in (1, 2, 3, 4, 5, 6):
        for trial in range(3000):
            c = rng.normal(size=deg + 1) * 10.0 ** rng.uniform(-2, 3, size=deg + 1)
            if trial % 7 == 0 and deg > 1:
                c[rng.integers(1, deg)] = 0.0
            m, st = oracle_mod.matlab_roots(c)
            w = np.roots(c)
            assert st == 0 and m.size == w.size
            total += 1
            if np.max(np.abs(m - w) / np.abs(w)) > 1e-8:
                # same multiset in another order: only where eigenvalue groups of (nearly) equal modulus deflate in an order
                # that hangs on the last bit (LAPACK builds differ among themselves there)
                assert np.max(np.abs(np.sort_complex(m) - np.sort_complex(w)) / np.abs(np.sort_complex(w))) < 1e-8
                swapped += 1
    assert swapped <= total // 500, (swapped, total)
    # leading zeros are stripped (fewer roots), trailing zeros become zero roots at the head of the result, NaN / Inf is an error
    m, st = oracle_mod.matlab_roots([0.0, 2.0, -6.0, 4.0])
    assert st == 0 and np.allclose(np.sort(m.real), [1.0, 2.0]) and m.size == 2
    m, st = oracle_mod.matlab_roots([1.0, -3.0, 2.0, 0.0, 0.0])
    assert st == 0 and m.size == 4 and np.all(m[:2] == 0) and np.allclose(np.sort(m[2:].real), [1.0, 2.0])
    assert oracle_mod.matlab_roots([1.0, np.nan, 2.0])[1] == 2


def test_matlab_unit_tables(oracle_mod, kat):
    """tests/unittests/testOptBraking.m, testOptSwitchTimes.m, testTimeScaling.m against the MATLAB-semantics twin (scenarios 10-12
    of the time-scaling table reach the polynomial cases, i.e. the positional root picks LTPlanner.m:346-416)."""
    m = kat["matlab_twins"]
    mm = m["opt_switch_times"]
    t_all = np.cumsum(np.array(mm["t_rel_rows"]).T, axis=1)
    for i in range(len(mm["q_goal"])):
        o = _matlab(oracle_mod, 1, 0.001, [mm["v_max"][i]], [mm["a_max"][i]], [mm["j_max"][i]])
        for sgn in ((1, -1) if i else (1,)):
            ok, t, d, mod = o.opt_switch_times(0, sgn * mm["q_goal"][i], sgn * mm["q_0"], sgn * mm["v_0"][i], sgn * mm["a_0"][i], mm["v_max"][i])
            assert ok and np.all(np.abs(t - t_all[i]) < mm["eps"]), (i, sgn, t)
    mm = m["time_scaling"]
    t_all = np.cumsum(np.array(mm["t_rel_rows"]).T, axis=1)
    cases = set()
    for i in range(len(mm["q_goal"])):
        o = _matlab(oracle_mod, 1, 0.001, [mm["v_max"][i]], [mm["a_max"][i]], [mm["j_max"][i]])
        for sgn in ((1, -1) if i else (1,)):
            ok, t, vd, mod, case = o.time_scaling(0, sgn * mm["q_goal"][i], sgn * mm["q_0"], sgn * mm["v_0"][i], sgn * mm["a_0"][i],
                                                  sgn * mm["dir"][i], t_all[i, -1])
            cases.add(case)
            assert np.all(np.abs(t - t_all[i]) < mm["eps"]), (i, sgn, case, t)
    assert cases - {1, 2}, "the table reaches polynomial cases"
    mm = m["opt_braking"]
    for i in range(len(mm["v_0"])):
        o = _matlab(oracle_mod, 1, 0.001, [mm["v_max"]], [mm["a_max"][i]], [mm["j_max"][i]])
        for sgn in ((1, -1) if i else (1,)):
            q, t, d = o.opt_braking(0, sgn * mm["v_0"][i], sgn * mm["a_0"][i])
            assert np.all(np.abs(t[:3] - np.array(mm["t_rel_rows"])[:, i]) < mm["eps"]) and abs(q - sgn * mm["q_goal"][i]) < mm["eps"]
    assert oracle_mod.lib().ltpo_matlab_flags(1) == 0


def test_matlab_grid_tests_and_readme_accuracy(oracle_mod, kat):
    L = oracle_mod.lib()
    out = (C.c_long * 6)(); worst = C.c_double(); total = C.c_double(); hist = (C.c_long * 9)()
    L.ltpo_kat_matlab_grid_one_joint(out, C.byref(worst), C.byref(total))
    success, not_finished, failure, time_error, scenarios, flags = list(out)
    assert scenarios == 101787 and success == scenarios and not_finished == failure == 0 and flags == 0      # gridTestOneJoint.m:72-74
    assert worst.value < 0.02
    L.ltpo_kat_matlab_grid_time_scaling(out, C.byref(worst), C.byref(total), hist)
    success, not_finished, failure, time_error, scenarios, flags = list(out)
    assert scenarios > 600000 and success == scenarios and not_finished == failure == time_error == 0 and flags == 0      # gridTestTimeScaling.m:95-97
    # README.md:128-136, "reproduced using tests/gridTestTimeScaling.m": average absolute error 0.003 rad, worst below 0.015 rad
    acc = kat["readme_accuracy"]
    assert round(total.value / scenarios, 3) == acc["mean_goal_error"] == 0.003
    assert worst.value < acc["worst_goal_error_below"]
    h = list(hist)
    assert sum(h) == scenarios and h[1] + h[2] > 0.99 * scenarios and sum(h[3:]) > 100        # the positional picks are exercised
    assert h[0] / scenarios < 1.2e-3                                                           # README.md:117-120 "less than 1 out of 1000" (measured 1.07)


def _numpy_get_trajectories(Ts, dof, j_max, t, dirv, mod, q0, v0, a0, vd):
    """An independent restatement of LTPlanner.m:486-625 in array form (1-based index arithmetic kept, arrays padded by one)."""
    L = int(np.max(np.ceil(t[:, 6] / Ts))) + 1
    jt = np.zeros((dof, L + 2))
    S = np.zeros((dof, 8), dtype=int)
    cv = np.zeros(dof, dtype=bool)

    def mod_(x, y):
        q = x / y
        n = np.round(q)
        if n != 0 and abs((q - n) / n) < np.finfo(float).eps:
            return 0.0
        return x - y * np.floor(q)
    for jn in range(dof):
        prof = np.array([-1, 0, 1, 0, -1, 0, 1] if mod[jn] else [1, 0, -1, 0, -1, 0, 1]) * dirv[jn] * j_max[jn]
        P = np.concatenate(([0.0], prof))                                  # 1-based
        fr = np.array([0.0] + [mod_(t[jn, k], Ts) for k in range(7)])
        s = S[jn]
        for k in range(1, 8):
            s[k] = int(np.floor(t[jn, k - 1] / Ts)) if k % 2 else int(np.ceil(t[jn, k - 1] / Ts))
        row = jt[jn]
        if s[1] > 0:
            row[1:s[1] + 1] = P[1]
        for k in range(2, 8):
            if s[k] - s[k - 1] > 0:
                row[s[k - 1] + 1:s[k] + 1] = P[k]
        if s[3] >= s[2]:
            row[s[1] + 1] += fr[1] / Ts * P[1]
            if s[2] > 0:
                row[s[2]] += (1 - fr[2] / Ts) * P[3]
            row[s[3] + 1] += fr[3] / Ts * P[3]
        elif s[2] > 0:
            row[s[2]] = row[s[2]] + fr[1] / Ts * P[1] + (fr[3] - fr[1]) / Ts * P[3]
        if s[4] > 0:
            row[s[4]] += (1 - fr[4] / Ts) * P[5]
        if s[3] - s[1] > 0:
            row[s[5] + 1] += fr[5] / Ts * P[5]
        elif s[5] > 0:
            row[s[5]] = row[s[5]] + fr[5] / Ts * P[5] + fr[1] / Ts * P[1] + (fr[3] - fr[1]) / Ts * P[3]
        if s[6] > 0:
            row[s[6]] += (1 - fr[6] / Ts) * P[7]
        row[s[7] + 1] += fr[7] / Ts * P[7]
        cv[jn] = s[4] - s[3] > 2
    jj = jt[:, 1:L + 1]
    a = Ts * np.cumsum(jj, axis=1) + np.asarray(a0)[:, None]
    a[dof - 1, S[dof - 1, 7]:] = 0.0                                        # :607 after the loop: the last joint only
    v = Ts * np.cumsum(a, axis=1) + np.asarray(v0)[:, None]
    for jn in range(dof):
        if cv[jn]:
            v[jn, S[jn, 3]:S[jn, 4] - 1] = vd[jn] * dirv[jn]
        v[jn, S[jn, 7]:] = 0.0
    q = Ts * np.cumsum(v, axis=1) + np.asarray(q0)[:, None]
    return L, q, v, a, jj


@pytest.mark.parametrize("dof,ts", [(1, 0.004), (3, 0.001), (6, 0.002)])
def test_matlab_sampler_against_a_numpy_restatement(oracle_mod, dof, ts):
    rng = np.random.default_rng(5 + dof)
    v_max, a_max, j_max = rng.uniform(0.8, 2.0, dof), rng.uniform(1.5, 6.0, dof), rng.uniform(10.0, 60.0, dof)
    o = _matlab(oracle_mod, dof, ts, v_max, a_max, j_max)
    c = oracle_mod.Oracle(dof, ts, [-9.0] * dof, [9.0] * dof, v_max, a_max, j_max)
    differs = 0
    for trial in range(60):
        qg, q0 = rng.uniform(-3, 3, dof), rng.uniform(-3, 3, dof)
        v0 = rng.uniform(-0.7, 0.7, dof) * v_max
        a0 = rng.uniform(-0.5, 0.5, dof) * np.minimum(a_max, np.sqrt(2 * j_max * (v_max - np.abs(v0))))
        r = o.plan_batch(qg, q0, v0, a0, sample=False)
        if r["status"][0] == 0:
            continue
        t, d, md, vd = r["t_scaled"][0], r["dir"][0], r["mod"][0], r["v_drive"][0]
        n, q, v, a, j = o.get_trajectory(t, d, md, q0, v0, a0, vd)
        L, qn, vn, an, jn = _numpy_get_trajectories(ts, dof, j_max, t, d, md, q0, v0, a0, vd)
        assert n == L == r["traj_len"][0]
        for got, want in ((q, qn), (v, vn), (a, an), (j, jn)):
            assert np.max(np.abs(got - want)) < 1e-11
        assert np.all(v[:, -1] == 0.0) and np.all(a[dof - 1, -1:] == 0.0)
        assert np.max(np.abs(q[:, -1] - qg)) < 0.03                       # the MATLAB sampler reaches the goal like the C++ one
        if dof > 1:
            # LTPlanner.m:607: the acceleration tail of every joint but the last keeps its residual (it is zeroed in the C++)
            s7 = np.floor(t[:, 6] / ts).astype(int)
            early = [i for i in range(dof - 1) if s7[i] + 3 < n]
            assert all(np.all(a[i, s7[i] + 1:] == a[i, s7[i] + 1]) for i in early)
        # and it differs from the C++ sampler in the samples around the switching times (corrections land one sample earlier)
        n2, q2, v2, a2, j2 = c.get_trajectory(t, d, md, q0, v0, a0, vd)
        differs += int(n2 == n and np.max(np.abs(j2 - j)) > 1e-6)
    assert differs > 20


def test_matlab_trajectory_semantics(oracle_mod):
    """LTPlanner.m:40-89 where it differs from cc:7-63: no position limits (q_0 outside [q_min, q_max] is planned, no end-limit
    verdict), the slowest joint's mod flag stays false, error() for velocity / acceleration violations."""
    D = 3
    v, a, j = [1.0, 1.5, 2.0], [2.0, 3.0, 4.0], [15.0, 20.0, 30.0]
    o = _matlab(oracle_mod, D, 0.002, v, a, j)
    c = oracle_mod.Oracle(D, 0.002, [-1.0] * D, [1.0] * D, v, a, j)
    qg, q0, z = np.array([[2.5, -2.0, 0.3]]), np.array([[1.7, 0.2, -0.4]]), np.zeros((1, D))
    rm, rc = o.plan_batch(qg, q0, z, z, sample=True), c.plan_batch(qg, q0, z, z, sample=True)
    assert rc["status"][0] == 0 and rm["status"][0] == 1 and rm["matlab_flags"][0] == 0 and rm["traj_len"][0] > 100
    bad = o.plan_batch(qg, q0, np.array([[0.0, 1.6, 0.0]]), z, sample=False)          # |v_0| > v_max: error() in LTPlanner.m:93-95
    assert bad["status"][0] == 0 and bad["matlab_flags"][0] & 2
    assert np.all(rm["mod"][0, rm["slowest"][0]] == 0)


def test_device_header_of_matlab_roots_equals_the_twin_on_the_host():
    """csrc/ltp_roots_matlab.hpp — the header the MATLAB-semantics kernels include: LAPACK's DGEEV path with the matrix in registers and
    every index a compile-time constant — compiled by plain g++ and compared with the twin's loop form (oracle/matlab_roots.inc) bit for
    bit: status, root count and every re / im entry on ~2.9 M polynomials of degree 0..6 (uniform and log-uniform coefficients, root
    clusters / exact multiples / near-real pairs that take both exceptional shifts, stripped leading / trailing zeros, magnitudes from
    subnormal to 2^1000, NaN / Inf). tests/test_gpu_matlab.py repeats the comparison on the device."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp"), "-s", "matlab_roots_test"])
    p = subprocess.run([os.path.join(root, "tests", "cpp", "matlab_roots_test"), "120", "20251005"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["mismatches"] == 0 and line["polynomials"] > 2_500_000
    assert line["with_both_exceptional_shifts"] > 10_000 and line["with_stripped_zeros"] > 50_000      # the rare paths were taken


def test_matlab_solver_stays_in_registers():
    """Build guard (hipcc cross-compiles here, no GPU): the MATLAB-semantics queue kernels keep the eigen-solve's matrix in registers. Inlined
    at its call sites the solver once spilled 483 registers per lane (1.7 KB of scratch), and with a dynamically indexed element it kept a
    matrix row in scratch memory (EXPERIMENTS E7.13): what is left per lane is the argument arrays of the out-of-line solver (c[7], re[6],
    im[6]) and a call frame."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kres.py"), "ltp_stage_kernels", "slow<"], capture_output=True, text=True, timeout=900).stdout
    seen = 0
    for line in out.splitlines():
        m = re.match(r"void ltp::(k_\w+)<(\d)>\s+VGPR\s+(\d+).*scratch\s+(\d+)\s+vspill\s+(\d+).*LDS\s+(\d+)", line)
        if m and int(m.group(2)) & 1:                       # SEM bit 0: MATLAB semantics
            seen += 1
            assert int(m.group(4)) <= 400 and int(m.group(5)) <= 16, line
            assert int(m.group(6)) <= 8192, line            # (and no 144 KB of matrices in LDS)
    assert seen == 4, out
