"""the MATLAB-semantics mode on the device (ltp_set_semantics(p, LTP_SEMANTICS_MATLAB)) against its CPU twin
(oracle.Oracle(..., semantics="matlab"), itself pinned by the MATLAB unit tables, the MATLAB grid tests and numpy.roots in
tests/test_matlab_twin.py) — bit-exact on integers and flags, 1e-9 on times and samples, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def amd():
    import longtermplanner_amd
    return longtermplanner_amd


def _pair(amd, oracle_mod, name, ts=0.001):
    D, lim = amd.limit_set(name)
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    ltp.setSemantics("matlab")
    orc = oracle_mod.Oracle(D, ts, semantics="matlab", **lim)
    return D, lim, ltp, orc


def test_matlab_roots_on_the_device(amd, oracle_mod):
    """ltp_roots_matlab.hpp vs oracle/matlab_roots.inc: the same restated LAPACK path -> same order, same classification."""
    D, lim = amd.limit_set("ref")
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    rng = np.random.default_rng(3)
    rows = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "planner_polynomials.npz"))["rows"]
    for deg in (1, 2, 3, 4, 5, 6):
        polys = [r[1:2 + deg] for r in rows if int(r[0]) == deg]
        for trial in range(4000):
            c = rng.normal(size=deg + 1) * 10.0 ** rng.uniform(-2, 3, size=deg + 1)
            if trial % 7 == 0 and deg > 1:
                c[rng.integers(1, deg)] = 0.0
            if trial % 11 == 0 and deg > 2:
                c[-1] = 0.0                                   # a zero root, placed first
            if trial % 13 == 0 and deg > 2:
                c[0] = 0.0                                    # a stripped leading coefficient: fewer roots
            polys.append(c)
        polys = np.array(polys)
        got, nr, st = ltp.matlabRoots(polys)
        worst = 0.0
        for i, c in enumerate(polys):
            want, wst = oracle_mod.matlab_roots(c)
            assert st[i] == wst and nr[i] == want.size, (deg, i)
            g = got[i, :nr[i]]
            assert np.array_equal(g.imag == 0, want.imag == 0), (deg, i)
            den = np.maximum(np.abs(want), 1e-300)
            worst = max(worst, float(np.max(np.abs(g - want) / den)) if want.size else 0.0)
        assert worst < 1e-12, (deg, worst)
    bad, nr, st = ltp.matlabRoots(np.array([[1.0, np.nan, 2.0, 3.0]]))
    assert st[0] == 2 and nr[0] == 0


def test_matlab_roots_on_the_device_bit_for_bit(amd, oracle_mod):
    """The register-resident solver (round 5: matrix in VGPRs, every index a constant, one out-of-line function per degree) against the
    twin's loop form, BIT for bit — status, root count, every re / im entry: ~160 k polynomials of degree 1..6 with uniform and
    log-uniform coefficients, root clusters and exact multiples (the exceptional shifts of iterations 10 and 20), stripped leading /
    trailing zeros, magnitudes from 2^-1000 to 2^1000, NaN / Inf coefficients. (The same header against the same twin on the host,
    2.9 M polynomials: tests/test_matlab_twin.py.)"""
    D, lim = amd.limit_set("ref")
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    rng = np.random.default_rng(20251005)
    n = 6000
    checked = noconv = 0
    rows = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "planner_polynomials.npz"))["rows"]
    for deg in (1, 2, 3, 4, 5, 6):
        planner = np.array([r[1:2 + deg] for r in rows if int(r[0]) == deg]).reshape(-1, deg + 1)      # what the planner really solves
        sets = [planner, rng.uniform(-1, 1, size=(n, deg + 1)),
                rng.choice([-1.0, 1.0], size=(n, deg + 1)) * np.exp(rng.uniform(-27.6, 27.6, size=(n, deg + 1)))]
        # from chosen roots: a cluster of width `spread` around `centre`, exact multiples, a few roots elsewhere
        centre = rng.uniform(-3, 3, size=(n, 1)); spread = np.exp(-rng.uniform(0, 30, size=(n, 1)))
        pick = rng.integers(0, 4, size=(n, deg))
        roots = np.where(pick == 0, centre, np.where(pick == 1, centre + spread * rng.uniform(-1, 1, size=(n, deg)), rng.uniform(-10, 10, size=(n, deg))))
        c = np.ones((n, 1))
        for k in range(deg):
            c = np.concatenate([c, np.zeros((n, 1))], axis=1) - np.concatenate([np.zeros((n, 1)), c * roots[:, k:k + 1]], axis=1)
        sets.append(c * np.exp(rng.uniform(-10, 10, size=(n, 1))))
        z = rng.uniform(-1, 1, size=(n, deg + 1))
        lead0 = rng.integers(0, deg + 2, size=n); trail0 = rng.integers(0, deg + 2, size=n)
        col = np.arange(deg + 1)[None, :]
        z[col < lead0[:, None]] = 0.0
        z[col > deg - trail0[:, None]] = 0.0
        sets.append(z)
        sets.append(rng.uniform(-1, 1, size=(n // 2, deg + 1)) * np.exp2(rng.integers(-1000, 1000, size=(n // 2, deg + 1)).astype(np.float64)))
        bad = rng.uniform(-1, 1, size=(64, deg + 1))
        bad[np.arange(64), rng.integers(0, deg + 1, size=64)] = np.where(rng.integers(0, 2, size=64) == 1, np.nan, np.inf)
        sets.append(bad)
        polys = np.ascontiguousarray(np.concatenate(sets, axis=0))
        got, nr, st = ltp.matlabRoots(polys)
        gre = np.ascontiguousarray(got.real).view(np.uint64); gim = np.ascontiguousarray(got.imag).view(np.uint64)
        for i, c in enumerate(polys):
            want, wst = oracle_mod.matlab_roots(c)
            assert st[i] == wst and nr[i] == want.size, (deg, i, c.tolist())
            k = int(nr[i])
            wre = np.ascontiguousarray(want.real).view(np.uint64); wim = np.ascontiguousarray(want.imag).view(np.uint64)
            same = ((gre[i, :k] == wre) | (np.isnan(got.real[i, :k]) & np.isnan(want.real))) & ((gim[i, :k] == wim) | (np.isnan(got.imag[i, :k]) & np.isnan(want.imag)))
            assert bool(np.all(same)), (deg, i, c.tolist(), got[i, :k].tolist(), want.tolist())
            checked += 1
            noconv += int(wst == 1)
    assert checked > 160_000


def test_matlab_unit_tables_on_the_device(amd, kat):
    """tests/unittests/*.m through the one-lane entry points in MATLAB semantics (all seven switching times)."""
    m = kat["matlab_twins"]
    mm = m["opt_switch_times"]
    t_all = np.cumsum(np.array(mm["t_rel_rows"]).T, axis=1)
    for i in range(len(mm["q_goal"])):
        ltp = amd.LongTermPlanner(1, 0.001, [0.0], [0.0], [mm["v_max"][i]], [mm["a_max"][i]], [mm["j_max"][i]], device=0)
        ltp.setSemantics("matlab")
        for sgn in ((1, -1) if i else (1,)):
            ok, t, d, mod = ltp.optSwitchTimes(0, sgn * mm["q_goal"][i], sgn * mm["q_0"], sgn * mm["v_0"][i], sgn * mm["a_0"][i], mm["v_max"][i])
            assert ok and np.all(np.abs(t - t_all[i]) < mm["eps"]) and ltp.lastMatlabFlags() == 0, (i, sgn, t)
    mm = m["time_scaling"]
    t_all = np.cumsum(np.array(mm["t_rel_rows"]).T, axis=1)
    cases = set()
    for i in range(len(mm["q_goal"])):
        ltp = amd.LongTermPlanner(1, 0.001, [0.0], [0.0], [mm["v_max"][i]], [mm["a_max"][i]], [mm["j_max"][i]], device=0)
        ltp.setSemantics("matlab")
        for sgn in ((1, -1) if i else (1,)):
            ok, t, vd, mod, case = ltp.timeScaling(0, sgn * mm["q_goal"][i], sgn * mm["q_0"], sgn * mm["v_0"][i], sgn * mm["a_0"][i],
                                                   sgn * mm["dir"][i], t_all[i, -1])
            cases.add(case)
            assert np.all(np.abs(t - t_all[i]) < mm["eps"]), (i, sgn, case, t)
    assert cases - {0, 1, 2}, "polynomial cases (positional root picks) are reached"
    mm = m["opt_braking"]
    for i in range(len(mm["v_0"])):
        ltp = amd.LongTermPlanner(1, 0.001, [0.0], [0.0], [mm["v_max"]], [mm["a_max"][i]], [mm["j_max"][i]], device=0)
        ltp.setSemantics("matlab")
        for sgn in ((1, -1) if i else (1,)):
            ok, q, t, d = ltp.optBraking(0, sgn * mm["v_0"][i], sgn * mm["a_0"][i])
            assert np.all(np.abs(t[:3] - np.array(mm["t_rel_rows"])[:, i]) < mm["eps"]) and abs(q - sgn * mm["q_goal"][i]) < mm["eps"]


@pytest.mark.parametrize("name,n,ts", [("panda", 200_000, 0.001), ("ref", 200_000, 0.001), ("ref30", 20_000, 0.001), ("ref", 50_000, 0.004)])
def test_matlab_records_parity(amd, oracle_mod, name, n, ts):
    from concurrent.futures import ThreadPoolExecutor
    D, lim, ltp, orc = _pair(amd, oracle_mod, name, ts)
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=909)
    q0[5] = q0[5] + 50.0                  # far outside [q_min, q_max]: LTPlanner.m has no position limits, the query is planned
    v0[9, 0] = 2.0 * lim["v_max"][0]      # LTPlanner.m:93-95 error()
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=False)
    parts = 16
    with ThreadPoolExecutor(parts) as ex:
        outs = list(ex.map(lambda i: orc.plan_batch(qg[i::parts], q0[i::parts], v0[i::parts], a0[i::parts], sample=False), range(parts)))
    o = {k: np.empty_like(outs[0][k], shape=(n,) + outs[0][k].shape[1:]) for k in outs[0] if isinstance(outs[0][k], np.ndarray)}
    for i, part in enumerate(outs):
        for k in o:
            o[k][i::parts] = part[k]
    planned = o["status"] != 0
    assert planned[5] and not planned[9] and (o["matlab_flags"][9] & 2)
    dev_planned = (r["status"] & ~amd.STATUS_MATLAB_COMPLEX) == 0
    assert np.array_equal(dev_planned, planned)
    assert np.array_equal((r["status"] & amd.STATUS_MATLAB_ERROR) != 0, (o["matlab_flags"] & 2) != 0)
    assert np.array_equal((r["status"] & amd.STATUS_MATLAB_COMPLEX) != 0, (o["matlab_flags"] & 1) != 0)
    assert not np.any(r["status"] & amd.STATUS_END_LIMIT)
    for k in ("slowest", "traj_len", "mod", "dir"):
        assert np.array_equal(r[k][planned], o[k][planned]), k
    worst = 0.0
    for k in ("t_opt", "t_scaled", "v_drive", "t_required"):
        a, b = r[k][planned], o[k][planned]
        d = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, np.abs(a - b))
        assert np.all(np.isfinite(d)), k
        worst = max(worst, float(d.max()))
    print(f"MATLAB semantics {name}: {int(planned.sum())} plans, worst |dt| {worst:.3e}, complex flags {int(np.sum(o['matlab_flags'] & 1))}, "
          f"errors {int(np.sum((o['matlab_flags'] & 2) != 0))}, mod joints {float(np.mean(o['mod'][planned])):.3f}")
    assert worst <= TOL
    # the two semantics really differ on this batch: other switching times for some joints
    c = amd.LongTermPlanner(D, ts, device=0, **lim).planBatchHost(qg, q0, v0, a0, sample=False)
    both = planned & ((c["status"] & 7) == 0)
    assert np.any(np.abs(c["t_scaled"][both] - r["t_scaled"][both]) > 1e-6) or name == "panda"


@pytest.mark.parametrize("name", ["panda", "ref", "ref30"])
def test_matlab_queue_spreading_batch_sizes(amd, oracle_mod, name):
    """Round 5: in MATLAB semantics the two queue kernels deal their queued lanes to all the blocks that run at once (1 .. 64 lanes per
    block by the queue's length, ltp_stage_kernels.hip slow_lanes_per_block). Batches whose queues hold fewer lanes than there are
    blocks, exactly a block's worth, and many rounds' worth: every record entry the twin's BITS (the arithmetic does not depend on
    which block a lane lands in)."""
    D, lim, ltp, orc = _pair(amd, oracle_mod, name)
    sizes = (1, 2, 9, 64, 65, 1000, 4097, 70001) if D == 7 else (1, 3, 64, 513, 9001)
    for n in sizes:
        qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=4040 + n)
        r = ltp.planBatchHost(qg, q0, v0, a0, sample=False)
        o = orc.plan_batch(qg, q0, v0, a0, sample=False)
        planned = o["status"] != 0
        assert np.array_equal((r["status"] & ~amd.STATUS_MATLAB_COMPLEX) == 0, planned), n
        for k in ("slowest", "traj_len", "mod", "dir"):
            assert np.array_equal(r[k][planned], o[k][planned]), (n, k)
        for k in ("t_opt", "t_scaled", "v_drive", "t_required"):
            a, b = np.ascontiguousarray(r[k][planned]), np.ascontiguousarray(o[k][planned])
            same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
            assert bool(np.all(same)), (n, k, float(np.nanmax(np.abs(a - b))))


@pytest.mark.parametrize("name,n,ts", [("panda", 600, 0.001), ("ref", 300, 0.001), ("ref30", 40, 0.001), ("ref", 400, 0.004)])
def test_matlab_dense_trajectory_parity(amd, oracle_mod, name, n, ts):
    import torch
    D, lim, ltp, orc = _pair(amd, oracle_mod, name, ts)
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=31337)
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    assert np.array_equal(r["traj_len"], o["traj_len"])
    worst = {k: 0.0 for k in "qvaj"}
    for p in range(n):
        if o["status"][p] == 0:
            continue
        L, q, v, a, j = orc.get_trajectory(o["t_scaled"][p], o["dir"][p], o["mod"][p], q0[p], v0[p], a0[p], o["v_drive"][p])
        g = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), D, L)
        for key, got, want in zip("qvaj", g, (q, v, a, j)):
            worst[key] = max(worst[key], float(np.max(np.abs(got - want))))
    print(f"MATLAB semantics dense {name}: {worst}")
    # j: a last-bit difference of a switching time reaches the fractional jerk sample times j_max / Ts
    assert worst["q"] <= TOL and worst["v"] <= TOL and worst["a"] <= 1e-8 and worst["j"] <= 1e-5
    # the consumers share the tables: restart states and envelopes are the bits of the rows
    dq = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    b = ltp.planSwitchTimesBatch(*dq)
    k = 37
    sq, sv, sa = ltp.stateAt(b, 0, n, k)
    env = ltp.envelopeBatch(b, 0, n, 64, 8).cpu().numpy()
    torch.cuda.synchronize()
    for p in range(0, n, max(1, n // 60)):
        L = int(r["traj_len"][p])
        if L <= 0:
            continue
        g = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), D, L)
        kk = min(k, L - 1)
        assert np.array_equal(sq[p].cpu().numpy(), g[0][:, kk]) and np.array_equal(sv[p].cpu().numpy(), g[1][:, kk])
        assert np.array_equal(sa[p].cpu().numpy(), g[2][:, kk])
        for w in range(8):
            lo, hi = min(w * 64, L - 1), min(w * 64 + 64, L)
            seg = g[0][:, lo:max(hi, lo + 1)]
            assert np.array_equal(env[p, :, w, 0], seg.min(axis=1)) and np.array_equal(env[p, :, w, 1], seg.max(axis=1))
    # float32 rows and capped rows go through the same tables
    ltp.setMaxSamples(100)
    m = min(n, 50)
    r2 = ltp.planBatchHost(qg[:m], q0[:m], v0[:m], a0[:m], sample=True)
    for p in range(m):
        L = int(r["traj_len"][p])
        if L <= 0:
            continue
        S = min(L, 100)
        g = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), D, L)
        g2 = amd.unpack_trajectory(r2["packed"], int(r2["offsets"][p]), D, S)
        for x in range(4):
            assert np.array_equal(g2[x], g[x][:, :S])


def test_matlab_mode_single_call_and_misuse(amd, oracle_mod):
    D, lim, ltp, orc = _pair(amd, oracle_mod, "ref", 0.004)
    qg, q0, v0, a0 = amd.generate_queries(4, lim, seed=5)
    tr = amd.Trajectory()
    ok = ltp.planTrajectory(qg[0], q0[0], v0[0], a0[0], tr)
    o = orc.plan_trajectory(qg[0], q0[0], v0[0], a0[0])
    assert ok and tr.length == o["length"]
    for got, want in ((tr.q, o["q"]), (tr.v, o["v"]), (tr.a, o["a"])):
        assert np.max(np.abs(np.asarray(got) - want)) <= TOL
    assert np.all(np.asarray(tr.v)[:, -1] == 0.0) and np.max(np.abs(np.asarray(tr.q)[:, -1] - qg[0])) < 0.03
    assert ltp.checkInputs(q0[0] + 100.0, v0[0], a0[0]) and not ltp.checkInputs(q0[0], v0[0] * 0 + 5.0, a0[0])
    # a batch planned with one semantics cannot be sampled with the other
    import torch
    dq = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    b = ltp.planSwitchTimesBatch(*dq)
    tile = torch.zeros(int(b.offsets[-1].item()) + 64, dtype=torch.float64, device="cuda")
    ltp.setSemantics("cpp")
    with pytest.raises(amd.LtpError):
        ltp.sampleBatch(b, 0, 4, tile)
    ltp.setSemantics("matlab")
    ltp.sampleBatch(b, 0, 4, tile)
    assert ltp.lastSamplerKernel().startswith("k_sample_walk_matlab")      # MATLAB semantics: the walk kernel for every row format ...
    tile2 = torch.zeros_like(tile)
    ltp.sampleBatch(b, 0, 4, tile2, walk=False)
    assert ltp.lastSamplerKernel().startswith("k_sample_tab")              # ... or the table pass (dof > 63, on request): same rows
    torch.cuda.synchronize()
    assert torch.equal(tile, tile2)


@pytest.mark.parametrize("limits,dof,n", [("panda", None, 900), ("ref", 9, 300), ("ref", 30, 80)])
def test_matlab_walk_sampler_equals_the_table_pass(amd, limits, dof, n):
    """MATLAB semantics through k_sample_walk_matlab_* (run tables built inside the sampler's block by for_each_run<MATLAB>) against
    the table pass (k_build_tables<MATLAB> + k_sample_tab*): every row format, bit for bit — rows, statuses, lengths."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ltp.setSemantics("matlab")
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=23))
    v0[3, 0] = 99.0                                              # (MATLAB's checkInputs has no position limits: a velocity it rejects)
    short = torch.arange(20, n, 7, device=qg.device)
    qg[short] = q0[short] + 0.03
    v0[short] = 0.0
    a0[short] = 0.0
    for cap, stride, f32 in ((0, 1, False), (0, 1, True), (0, 4, False), (256, 1, False), (64, 1, False), (100, 3, True), (16, 1, False), (2000, 2, False)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("walk", "tables"):
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            full = torch.full((int(b.offsets[-1].item()),), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **({} if mode == "walk" else dict(walk=False)))
            kern = ltp.lastSamplerKernel()
            assert kern.startswith("k_sample_walk_matlab" if mode == "walk" else "k_sample_tab"), (mode, kern)
            sub = torch.full((int((b.offsets[n - 2] - b.offsets[9]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 9, n - 11, sub, spread=5, **({} if mode == "walk" else dict(walk=False)))
            torch.cuda.synchronize()
            res[mode] = (full, sub, b.status.clone(), b.traj_len.clone())
        for k, (got, want) in enumerate(zip(res["walk"], res["tables"])):
            assert torch.equal(got, want), (cap, stride, f32, k)
        assert int((res["walk"][3] > 0).sum().item()) > 0.9 * n
