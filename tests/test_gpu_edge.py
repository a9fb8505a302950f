"""Edge cases of the batched path: empty / ragged batches, invalid and degenerate queries, tile overflow, layouts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(scope="module")
def amd():
    import longtermplanner_amd as m
    return m


@pytest.fixture(scope="module")
def ref7(amd, oracle_mod):
    D, lim = amd.limit_set("ref")
    return D, lim, amd.LongTermPlanner(D, 0.001, device=0, **lim), oracle_mod.Oracle(D, 0.001, **lim)


def _compare(amd, r, o, ltp_dof, q0, v0, a0, orc):
    ran = (r["status"] & 7) == 0
    assert np.array_equal(ran & (r["traj_len"] > 0), (o["status"] != 0) & (o["traj_len"] > 0))
    for p in np.nonzero(ran & (r["traj_len"] > 0))[0]:
        assert np.nanmax(np.abs(r["t_scaled"][p] - o["t_scaled"][p])) <= TOL
        L, q, v, a, j = orc.get_trajectory(o["t_scaled"][p], o["dir"][p], o["mod"][p], q0[p], v0[p], a0[p], o["v_drive"][p])
        g = amd.unpack_trajectory(r["packed"], int(r["offsets"][p]), ltp_dof, L)
        for got, want in zip(g, (q, v, a, j)):
            assert np.max(np.abs(got - want)) <= TOL


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 130])
def test_ragged_batch_sizes(amd, ref7, n):
    D, lim, ltp, orc = ref7
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=n + 1)
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    assert r["status"].shape == (n,) and r["offsets"].shape == (n + 1,)
    if n == 0:
        assert r["offsets"][0] == 0 and r["packed"].size == 0
        return
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    _compare(amd, r, o, D, q0, v0, a0, orc)


def test_invalid_queries_are_flagged_and_skipped(amd, ref7):
    D, lim, ltp, orc = ref7
    qg, q0, v0, a0 = amd.generate_queries(40, lim, seed=8)
    q0[3, 2] = 3.5            # outside q_max
    v0[7, 0] = -1.5           # |v| > v_max
    a0[11, 6] = 2.5           # |a| > a_max
    v0[13, 1], a0[13, 1] = 0.95, 1.9   # v + a|a|/(2j) > v_max
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    bad = [3, 7, 11, 13]
    for p in range(40):
        if p in bad:
            assert r["status"][p] & amd.STATUS_INVALID_INPUT
            assert r["traj_len"][p] == 0 and r["offsets"][p + 1] == r["offsets"][p]
        else:
            assert r["status"][p] & amd.STATUS_INVALID_INPUT == 0
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    assert np.array_equal(o["status"] == 0, np.isin(np.arange(40), bad))
    _compare(amd, r, o, D, q0, v0, a0, orc)
    assert ltp.checkInputs(q0[0], v0[0], a0[0]) and not ltp.checkInputs(q0[3], v0[3], a0[3])


def test_degenerate_plans(amd, oracle_mod):
    # the all-zero plan (traj_len 1, the reference writes out of range), dir == 0, and
    # switching times that are exact multiples of the sample time
    lim = dict(q_min=[-3.1], q_max=[3.1], v_max=[2.0], a_max=[2.0], j_max=[4.0])
    for ts in (0.001, 0.25, 0.5):
        ltp = amd.LongTermPlanner(1, ts, device=0, **lim)
        orc = oracle_mod.Oracle(1, ts, **lim)
        qg = np.array([[1.0], [1.0], [0.0], [-1.0], [0.5], [1.002]])
        q0 = np.array([[1.0], [0.0], [1.0], [1.0], [0.5], [1.0]])
        v0 = np.zeros((6, 1)); a0 = np.zeros((6, 1))
        r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
        o = orc.plan_batch(qg, q0, v0, a0, sample=False)
        assert np.array_equal(r["traj_len"], o["traj_len"])
        assert np.array_equal(r["dir"], o["dir"]) and r["dir"][0, 0] == 0.0
        _compare(amd, r, o, 1, q0, v0, a0, orc)
        if ts < 0.5:
            # (at Ts = 0.5 optBraking's "-t_sample" threshold, cc:685, keeps a phantom phase: traj_len 2 in the reference too)
            assert r["traj_len"][0] == 1
            q, v, a, j = amd.unpack_trajectory(r["packed"], int(r["offsets"][0]), 1, 1)
            assert (q[0, 0], v[0, 0], a[0, 0], j[0, 0]) == (1.0, 0.0, 0.0, 0.0)


def test_non_finite_inputs_do_not_fault(amd, ref7):
    D, lim, ltp, orc = ref7
    qg, q0, v0, a0 = amd.generate_queries(70, lim, seed=21)
    qg[5, 3] = np.nan
    q0[9, :] = np.nan
    v0[20, 2] = np.inf
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    sampled = ((r["status"] & 7) == 0) & (r["traj_len"] > 0)
    assert np.array_equal(sampled, (o["status"] != 0) & (o["traj_len"] > 0))
    assert not sampled[5] and not sampled[9] and not sampled[20]
    assert np.all(np.isfinite(r["packed"]))


def test_default_constructed_planner(amd):
    ltp = amd.LongTermPlanner(device=0)           # dof 0, Ts 0.001 (long_term_planner.h:103-105)
    traj = amd.Trajectory(length=-5)
    assert ltp.planTrajectory([], [], [], [], traj) is False and traj.length == -5
    assert ltp.checkInputs([], [], []) is True


def test_end_limit_failures_keep_the_trajectory(amd, ref7):
    # cc:59-61: planTrajectory returns false when the last sample is outside the limits, but the trajectory is
    # delivered. Such plans are ~0.1 % of random batches; find some with the oracle and check them on the device.
    D, lim, ltp, orc = ref7
    lim4 = dict(lim)
    ltp4 = amd.LongTermPlanner(D, 0.004, device=0, **lim4)
    import oracle
    orc4 = oracle.Oracle(D, 0.004, **lim4)
    qg, q0, v0, a0 = amd.generate_queries(20000, lim, seed=77)
    o = orc4.plan_batch(qg, q0, v0, a0, sample=True)
    idx = np.nonzero(o["status"] == 2)[0]
    assert idx.size >= 3
    r = ltp4.planBatchHost(qg[idx], q0[idx], v0[idx], a0[idx], sample=True)
    assert np.all(r["status"] == amd.STATUS_END_LIMIT)
    assert np.array_equal(r["traj_len"], o["traj_len"][idx]) and np.all(r["traj_len"] > 1)
    traj = amd.Trajectory()
    assert ltp4.planTrajectory(qg[idx[0]], q0[idx[0]], v0[idx[0]], a0[idx[0]], traj) is False and traj.length == r["traj_len"][0]


def test_layouts_chunking_overflow_and_determinism(amd, ref7):
    import torch
    D, lim, ltp, orc = ref7
    n = 3000
    qm = ltp.generateQueries(n, seed=5)
    jm = [x.t().contiguous() for x in qm]
    b1 = ltp.planSwitchTimesBatch(*qm, layout="query_major")
    b2 = ltp.planSwitchTimesBatch(*jm, layout="joint_major")
    torch.cuda.synchronize()
    for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status", "offsets"):
        assert torch.equal(getattr(b1, k), getattr(b2, k)), k
    off = b1.offsets.cpu().numpy().view(np.uint64)
    total = int(off[-1])
    full = torch.zeros(total, dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b1, 0, n, full)
    again = torch.zeros(total, dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b2, 0, n, again, streaming=False, spread=1)
    torch.cuda.synchronize()
    assert torch.equal(full, again), "sampling is not deterministic across layouts / store flavours / block orders"
    # chunked into a small reused tile == unchunked
    cap = int(off[700] - off[0]) + 64
    tile = torch.zeros(cap, dtype=torch.float64, device="cuda")
    first = 0
    while first < n:
        end = int(np.searchsorted(off, off[first] + np.uint64(cap), side="right")) - 1
        assert end > first
        tile.zero_()
        ltp.sampleBatch(b1, first, end - first, tile)
        torch.cuda.synchronize()
        assert torch.equal(tile[: int(off[end] - off[first])], full[int(off[first]): int(off[end])])
        first = end
    # a tile that is too small: plans that do not fit are flagged, nothing is written out of bounds
    guard = torch.full((cap + 4096,), 7.0, dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b1, 0, n, guard[:cap])
    torch.cuda.synchronize()
    st = b1.status.cpu().numpy()
    fits = (off[1:] - off[0]) <= cap
    assert np.array_equal((st & amd.STATUS_OVERFLOW) != 0, ~fits & (b1.traj_len.cpu().numpy() > 0))
    assert torch.all(guard[cap:] == 7.0)


def test_first_n_samples_and_replanning(amd, ref7, oracle_mod):
    # "first N samples only" rows and the on-device receding-horizon gather
    import torch
    D, lim, ltp, orc = ref7
    n, cap, k = 500, 200, 150
    qm = ltp.generateQueries(n, seed=9)
    b = ltp.planSwitchTimesBatch(*qm)
    torch.cuda.synchronize()
    off_full = b.offsets.cpu().numpy().view(np.uint64).copy()
    full = torch.zeros(int(off_full[-1]), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, 0, n, full)
    torch.cuda.synchronize()
    status_full = b.status.cpu().numpy().copy()
    lens = b.traj_len.cpu().numpy()
    try:
        ltp.setMaxSamples(cap)
        b2 = ltp.planSwitchTimesBatch(*qm)
        torch.cuda.synchronize()
        assert torch.equal(b2.traj_len, b.traj_len), "traj_len keeps the reference's Trajectory::length"
        off_cap = b2.offsets.cpu().numpy().view(np.uint64)
        stored = np.minimum(lens, cap)
        assert np.array_equal(np.diff(off_cap.astype(np.int64)), 4 * D * ((stored + 31) // 32 * 32) * (lens > 0))
        capped = torch.zeros(int(off_cap[-1]), dtype=torch.float64, device="cuda")
        ltp.sampleBatch(b2, 0, n, capped)
        torch.cuda.synchronize()
        assert np.array_equal(b2.status.cpu().numpy(), status_full), "the end-limit check still sees the whole trajectory"
        hf, hc = full.cpu().numpy(), capped.cpu().numpy()
        for p in range(0, n, 7):
            if lens[p] <= 0:
                continue
            want = amd.unpack_trajectory(hf, int(off_full[p]), D, int(lens[p]))
            got = amd.unpack_trajectory(hc, int(off_cap[p]), D, int(stored[p]))
            for w, g in zip(want, got):
                assert np.array_equal(w[:, : stored[p]], g)
        # receding horizon: state at sample k of every capped trajectory, gathered on the device
        q1, v1, a1 = ltp.replanStates(b2, 0, n, capped, k)
        idx = torch.randint(0, cap, (n,), dtype=torch.int32, device="cuda")
        q2, v2, a2 = ltp.replanStates(b2, 0, n, capped, idx, layout="joint_major")
        torch.cuda.synchronize()
        hi = idx.cpu().numpy()
        for p in range(0, n, 5):
            if lens[p] <= 0:
                for t1, src in ((q1, qm[1]), (v1, qm[2]), (a1, qm[3])):
                    assert torch.equal(t1[p], src[p])
                continue
            q, v, a, _ = amd.unpack_trajectory(hc, int(off_cap[p]), D, int(stored[p]))
            kk = min(k, stored[p] - 1)
            assert np.array_equal(q1[p].cpu().numpy(), q[:, kk]) and np.array_equal(v1[p].cpu().numpy(), v[:, kk])
            assert np.array_equal(a1[p].cpu().numpy(), a[:, kk])
            k2 = min(int(hi[p]), stored[p] - 1)
            assert np.array_equal(q2[:, p].cpu().numpy(), q[:, k2]) and np.array_equal(a2[:, p].cpu().numpy(), a[:, k2])
        # and the replanned batch agrees with the oracle replanning from ITS trajectory's sample k
        b3 = ltp.planSwitchTimesBatch(qm[0], q1, v1, a1)
        torch.cuda.synchronize()
        host = [x.cpu().numpy() for x in qm]
        o = orc.plan_batch(*host, sample=False)
        for p in range(0, n, 25):
            if o["status"][p] == 0:
                continue
            L, q, v, a, j = orc.get_trajectory(o["t_scaled"][p], o["dir"][p], o["mod"][p], host[1][p], host[2][p], host[3][p], o["v_drive"][p])
            kk = min(k, min(L, cap) - 1)
            o2 = orc.plan_batch(host[0][p], q[:, kk], v[:, kk], a[:, kk], sample=False)
            if o2["status"][0] == 0:
                assert b3.status[p].item() & 7
                continue
            assert np.max(np.abs(b3.t_scaled[p].cpu().numpy() - o2["t_scaled"][0])) <= TOL
    finally:
        ltp.setMaxSamples(0)


def test_float32_rows_are_the_rounded_float64_rows(amd, ref7):
    # float rows = the same binary64 results rounded once; offsets are format-independent
    import torch
    D, lim, ltp, orc = ref7
    n = 400
    qm = ltp.generateQueries(n, seed=17)
    b = ltp.planSwitchTimesBatch(*qm)
    torch.cuda.synchronize()
    total = int(b.offsets[-1].item())
    f64 = torch.zeros(total, dtype=torch.float64, device="cuda")
    f32 = torch.zeros(total, dtype=torch.float32, device="cuda")
    ltp.sampleBatch(b, 0, n, f64)
    st64 = b.status.clone()
    ltp.sampleBatch(b, 0, n, f32)
    torch.cuda.synchronize()
    assert torch.equal(b.status, st64)
    assert torch.equal(f32, f64.to(torch.float32)), "float rows must be the double rows rounded to nearest"
    q1, v1, a1 = ltp.replanStates(b, 0, n, f32, 40)
    q2, v2, a2 = ltp.replanStates(b, 0, n, f64, 40)
    torch.cuda.synchronize()
    live = (b.traj_len > 0)
    assert torch.equal(q1[live], q2[live].to(torch.float32).to(torch.float64)) and torch.equal(a1[live], a2[live].to(torch.float32).to(torch.float64))


@pytest.mark.parametrize("stride,cap", [(4, 0), (7, 50), (1000000, 0)])
def test_strided_rows_are_a_decimation_of_the_full_rows(amd, ref7, stride, cap):
    # rows that store samples 0, stride, 2*stride, ... (optionally capped) hold exactly those samples
    import torch
    D, lim, ltp, orc = ref7
    n = 300
    qm = ltp.generateQueries(n, seed=23)
    b = ltp.planSwitchTimesBatch(*qm)
    torch.cuda.synchronize()
    off_full = b.offsets.cpu().numpy().view(np.uint64).copy()
    full = torch.zeros(int(off_full[-1]), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, 0, n, full)
    torch.cuda.synchronize()
    status_full = b.status.cpu().numpy().copy()
    lens = b.traj_len.cpu().numpy()
    hf = full.cpu().numpy()
    try:
        ltp.setSampleStride(stride)
        ltp.setMaxSamples(cap)
        b2 = ltp.planSwitchTimesBatch(*qm)
        torch.cuda.synchronize()
        off = b2.offsets.cpu().numpy().view(np.uint64)
        for dtype in (torch.float64, torch.float32):
            dec = torch.zeros(int(off[-1]), dtype=dtype, device="cuda")
            ltp.sampleBatch(b2, 0, n, dec)
            torch.cuda.synchronize()
            assert np.array_equal(b2.status.cpu().numpy(), status_full)
            hd = dec.cpu().numpy()
            for p in range(0, n, 3):
                if lens[p] <= 0:
                    continue
                stored = ltp.storedSamples(int(lens[p]))
                want_cnt = -(-int(lens[p]) // stride)
                assert stored == (min(want_cnt, cap) if cap else want_cnt)
                want = amd.unpack_trajectory(hf, int(off_full[p]), D, int(lens[p]))
                got = amd.unpack_trajectory(hd, int(off[p]), D, stored)
                for w, g in zip(want, got):
                    ref = w[:, ::stride][:, :stored]
                    assert np.array_equal(ref.astype(hd.dtype), g)
    finally:
        ltp.setSampleStride(1)
        ltp.setMaxSamples(0)


def test_goal_precheck_is_opt_in_and_changes_nothing_else(amd, ref7):
    """with setGoalCheck a q_goal outside [q_min, q_max] is rejected up front (status 64); by default
    the reference's behaviour stands (planned, sampled, END_LIMIT false, cc:59-61)."""
    D, lim, _, orc = ref7
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qg, q0, v0, a0 = amd.generate_queries(200, lim, seed=31)
    outside = [5, 17, 64, 199]
    qg[5, 0] = 3.3; qg[17, 6] = -3.2; qg[64, 3] = 4.0; qg[199, 2] = np.nan
    base = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    assert not np.any(base["status"] & amd.STATUS_GOAL_OUTSIDE)
    for p in outside[:3]:
        assert base["status"][p] == amd.STATUS_END_LIMIT and base["traj_len"][p] > 0     # reference behaviour
    ltp.setGoalCheck(True)
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    for p in range(200):
        if p in outside:
            assert r["status"][p] & amd.STATUS_GOAL_OUTSIDE
            assert r["traj_len"][p] == 0 and r["offsets"][p + 1] == r["offsets"][p]
        else:
            assert r["status"][p] == base["status"][p] and r["traj_len"][p] == base["traj_len"][p]
            assert np.array_equal(r["t_scaled"][p], base["t_scaled"][p], equal_nan=True)
            a = r["packed"][int(r["offsets"][p]):int(r["offsets"][p + 1])]
            b = base["packed"][int(base["offsets"][p]):int(base["offsets"][p + 1])]
            assert np.array_equal(a, b)
    # single-call API: false, trajectory untouched
    t = amd.Trajectory()
    assert ltp.planTrajectory(qg[5], q0[5], v0[5], a0[5], t) is False and t.length == 0
    ltp.setGoalCheck(False)
    assert np.array_equal(ltp.planBatchHost(qg, q0, v0, a0, sample=True)["status"], base["status"])


@pytest.mark.parametrize("limits,n,window,n_windows", [("ref", 700, 64, 130), ("ref", 300, 1000, 4), ("panda", 500, 7, 300),
                                                        ("ref30", 60, 256, 33), ("ref", 200, 1, 40)])
def test_envelope_consumer_equals_reduced_dense_rows(amd, oracle_mod, limits, n, window, n_windows):
    """SURVEY §8(f).2 on-device consumer: ltp_envelope_batch == min / max over windows of the rows ltp_sample_batch
    stores (bit for bit), and within 1e-9 of the same reduction of the oracle's trajectories."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    orc = oracle_mod.Oracle(D, 0.001, **lim)
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=77)
    q0[3, 0] = 99.0                                            # one rejected plan -> NaN envelope
    dev = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    b = ltp.planSwitchTimesBatch(*dev)
    first, count = 2, n - 5                                    # a sub-range, like a chunk
    guard = torch.full((count * D * n_windows * 2 + 64,), 7.0, dtype=torch.float64, device="cuda")
    env = ltp.envelopeBatch(b, first, count, window, n_windows, out=guard[: count * D * n_windows * 2])
    off = b.offsets.cpu().numpy().view(np.uint64)
    tile = torch.zeros(int(off[first + count] - off[first]), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, first, count, tile)
    torch.cuda.synchronize()
    assert torch.all(guard[count * D * n_windows * 2:] == 7.0)
    env = env.cpu().numpy().reshape(count, D, n_windows, 2)
    tile = tile.cpu().numpy()
    lens = b.traj_len.cpu().numpy()
    status = b.status.cpu().numpy()
    assert status[3] & amd.STATUS_INVALID_INPUT and np.all(np.isnan(env[3 - first]))
    worst = 0.0
    for i in range(count):
        p = first + i
        L = int(lens[p])
        if L == 0:
            assert np.all(np.isnan(env[i]))
            continue
        q = amd.unpack_trajectory(tile, int(off[p] - off[first]), D, L)[0]            # [D][L] as stored by the sampler
        pad = np.concatenate([q, np.repeat(q[:, -1:], max(window * n_windows - L, 0) + window, axis=1)], axis=1)
        win = pad[:, : window * n_windows].reshape(D, n_windows, window)
        # windows that start inside the trajectory but run past its end only see the samples that exist, which the
        # padding with the last sample does not change; windows past the end hold the last sample
        assert np.array_equal(env[i, :, :, 0], win.min(axis=2)), (p, L)
        assert np.array_equal(env[i, :, :, 1], win.max(axis=2)), (p, L)
        if i % 16 == 0:
            o = orc.plan_trajectory(qg[p], q0[p], v0[p], a0[p])
            assert o["length"] == L
            qo = o["q"]
            pad = np.concatenate([qo, np.repeat(qo[:, -1:], max(window * n_windows - L, 0) + window, axis=1)], axis=1)
            wo = pad[:, : window * n_windows].reshape(D, n_windows, window)
            worst = max(worst, float(np.max(np.abs(env[i, :, :, 0] - wo.min(axis=2)))), float(np.max(np.abs(env[i, :, :, 1] - wo.max(axis=2)))))
    assert worst <= TOL
    # the consumer applies the end-limit check too (cc:59-61), like the sampler
    b2 = ltp.planSwitchTimesBatch(*dev)
    ltp.envelopeBatch(b2, 0, n, window, n_windows)
    b3 = ltp.planSwitchTimesBatch(*dev)
    t3 = torch.zeros(int(off[n] - off[0]), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b3, 0, n, t3)
    torch.cuda.synchronize()
    assert torch.equal(b2.status, b3.status)


@pytest.mark.parametrize("cap", [40, 100, 300])
def test_short_rows_every_wave_mapping_and_both_row_types(amd, ref7, cap):
    """First-N-samples rows shorter than the block are shared out over the waves differently (1, 2 or 4 waves per
    row); every mapping, in float64 and float32, must give the leading samples of the full rows."""
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    n = 300
    qm = ltp.generateQueries(n, seed=23)
    b = ltp.planSwitchTimesBatch(*qm)
    total = int(b.offsets[-1].item())
    full = torch.zeros(total, dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, 0, n, full)
    torch.cuda.synchronize()
    off_full = b.offsets.cpu().numpy().view(np.uint64).copy()
    lens = b.traj_len.cpu().numpy()
    hf = full.cpu().numpy()
    ltp.setMaxSamples(cap)
    b2 = ltp.planSwitchTimesBatch(*qm)
    off = b2.offsets.cpu().numpy().view(np.uint64)
    for dtype in (torch.float64, torch.float32):
        guard = torch.full((int(off[-1]) + 256,), 7.0, dtype=dtype, device="cuda")
        ltp.sampleBatch(b2, 0, n, guard[: int(off[-1])])
        torch.cuda.synchronize()
        assert torch.all(guard[int(off[-1]):] == 7.0)
        hc = guard.cpu().numpy()
        for p in range(0, n, 3):
            if lens[p] <= 0:
                continue
            stored = min(int(lens[p]), cap)
            want = amd.unpack_trajectory(hf, int(off_full[p]), D, int(lens[p]))
            got = amd.unpack_trajectory(hc, int(off[p]), D, stored)
            for w, g in zip(want, got):
                assert np.array_equal(w[:, :stored].astype(hc.dtype), g), (p, cap, dtype)


def test_a_second_handle_can_sample_what_the_first_one_planned(amd, ref7):
    """One handle per stream is the documented pattern: a handle that never planned must be able to sample / reduce."""
    import torch
    D, lim, ltp, _ = ref7
    other = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qm = ltp.generateQueries(200, seed=3)
    b = ltp.planSwitchTimesBatch(*qm)
    total = int(b.offsets[-1].item())
    t1 = torch.zeros(total, dtype=torch.float64, device="cuda")
    t2 = torch.zeros(total, dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, 0, 200, t1)
    other.sampleBatch(b, 0, 200, t2)
    e1 = ltp.envelopeBatch(b, 0, 200, 32, 16)
    e2 = amd.LongTermPlanner(D, 0.001, device=0, **lim).envelopeBatch(b, 0, 200, 32, 16)
    torch.cuda.synchronize()
    assert torch.equal(t1, t2) and torch.equal(e1.nan_to_num(7.0), e2.nan_to_num(7.0))


def test_trajectory_longer_than_int_range_is_flagged_not_sampled(amd, oracle_mod):
    """DEFINED here (reference: undefined double -> int conversion): a plan whose length does not fit an int is
    treated like non-finite switching times — LTP_STATUS_NONFINITE, traj_len 0 — in the device path and the oracle."""
    D, lim = amd.limit_set("ref")
    ts = 1e-10                                                 # seconds-long motions -> > 2^31 samples
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    orc = oracle_mod.Oracle(D, ts, **lim)
    qg, q0, v0, a0 = amd.generate_queries(70, lim, seed=12)
    qg[5], v0[5], a0[5] = q0[5], 0.0, 0.0                      # an all-zero plan still has a (short) trajectory
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    assert np.array_equal(r["traj_len"], o["traj_len"])
    long_ = np.arange(70) != 5
    assert np.all(r["traj_len"][long_] == 0) and np.all(r["status"][long_] & amd.STATUS_NONFINITE)
    assert r["traj_len"][5] > 0 and r["status"][5] == 0
    assert np.nanmax(np.abs(r["t_scaled"] - o["t_scaled"])) <= TOL
    # the one-joint getTrajectory entry point defines it the same way
    g = ltp.getTrajectoryBatchHost(r["t_scaled"][:8], r["dir"][:8], r["mod"][:8], q0[:8], v0[:8], a0[:8], r["v_drive"][:8])
    assert np.all(g["traj_len"][[0, 1, 2, 3, 4, 6, 7]] == 0) and g["traj_len"][5] == r["traj_len"][5]


def test_two_handles_on_two_streams_run_concurrently(amd, ref7):
    """One handle per stream: concurrent batches on two streams give the results of serial runs."""
    import torch
    D, lim, _, _ = ref7
    planners = [amd.LongTermPlanner(D, 0.001, device=0, **lim) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    n = 6000
    queries = [planners[0].generateQueries(n, seed=100 + i) for i in range(2)]
    serial = []
    for i in range(2):
        b = planners[i].planSwitchTimesBatch(*queries[i])
        t = torch.zeros(int(b.offsets[-1].item()), dtype=torch.float64, device="cuda")
        planners[i].sampleBatch(b, 0, n, t)
        torch.cuda.synchronize()
        serial.append((b.t_scaled.clone(), b.status.clone(), t))
    torch.cuda.synchronize()
    out = [None, None]
    tiles = [torch.zeros_like(serial[i][2]) for i in range(2)]
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                b = planners[i].planSwitchTimesBatch(*queries[i])
                planners[i].sampleBatch(b, 0, n, tiles[i])
                out[i] = b
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(out[i].t_scaled, serial[i][0]) and torch.equal(out[i].status, serial[i][1])
        assert torch.equal(tiles[i], serial[i][2])


@pytest.mark.parametrize("limits,n", [("ref", 3000), ("panda", 3000), ("ref30", 300)])
def test_state_at_has_the_bits_of_the_sampled_rows(amd, limits, n):
    """ltp_state_at_batch (no rows) == the row elements ltp_sample_batch stores at sample k, for every k."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qm = ltp.generateQueries(n, seed=41)
    qm[1][7, 0] = 99.0                                          # a rejected plan keeps its start state
    b = ltp.planSwitchTimesBatch(*qm)
    tile = torch.zeros(int(b.offsets[-1].item()), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(b, 0, n, tile)
    lens = b.traj_len
    gen = torch.Generator(device="cuda").manual_seed(5)
    for mode in ("random", "first", "last", "beyond", "uniform"):
        if mode == "random":
            k = (torch.rand(n, device="cuda", generator=gen) * lens.clamp(min=1)).to(torch.int32)
        elif mode == "first":
            k = torch.zeros(n, dtype=torch.int32, device="cuda")
        elif mode == "last":
            k = (lens - 1).clamp(min=0).to(torch.int32)
        elif mode == "beyond":
            k = (lens + 1000).to(torch.int32)
        else:
            k = 137
        want = ltp.replanStates(b, 0, n, tile, k)
        got = ltp.stateAt(b, 0, n, k)
        got_jm = ltp.stateAt(b, 3, n - 5, k if isinstance(k, int) else k[3:n - 2].contiguous(), layout="joint_major")
        torch.cuda.synchronize()
        for w, g, gj in zip(want, got, got_jm):
            assert torch.equal(w, g), mode
            assert torch.equal(w[3:n - 2].t().contiguous(), gj), mode
    assert torch.equal(got[0][7], qm[1][7])


@pytest.mark.parametrize("dof", [17, 64])
def test_many_joints(amd, oracle_mod, dof):
    """dof well beyond one joint group (8): several rounds in the stage kernels, several sampler items per plan."""
    rng = np.random.default_rng(dof)
    lim = dict(q_min=-rng.uniform(1.0, 3.0, dof), q_max=rng.uniform(1.0, 3.0, dof), v_max=rng.uniform(0.5, 2.5, dof),
               a_max=rng.uniform(1.0, 10.0, dof), j_max=rng.uniform(10.0, 500.0, dof))
    ltp = amd.LongTermPlanner(dof, 0.002, device=0, **lim)
    orc = oracle_mod.Oracle(dof, 0.002, **lim)
    n = 150
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=dof)
    r = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    assert np.array_equal(r["slowest"], o["slowest"]) and np.array_equal(r["traj_len"], o["traj_len"])
    assert np.array_equal(r["mod"], o["mod"])
    _compare(amd, r, o, dof, q0, v0, a0, orc)


def test_envelope_through_the_host_pointer_api(amd, ref7):
    import torch
    D, lim, ltp, _ = ref7
    n, W, K = 500, 40, 50
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=55)
    q0[9, 3] = -77.0
    rec, env = ltp.planEnvelopeHost(qg, q0, v0, a0, W, K)
    dev = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    b = ltp.planSwitchTimesBatch(*dev)
    want = ltp.envelopeBatch(b, 0, n, W, K)
    torch.cuda.synchronize()
    assert np.array_equal(env, want.cpu().numpy(), equal_nan=True)
    assert np.array_equal(rec["status"], b.status.cpu().numpy()) and np.array_equal(rec["traj_len"], b.traj_len.cpu().numpy())
    assert np.array_equal(rec["t_scaled"], b.t_scaled.cpu().numpy(), equal_nan=True)
    assert rec["status"][9] & amd.STATUS_INVALID_INPUT and np.all(np.isnan(env[9]))


def test_replan_states_skips_plans_the_sampler_skipped(amd, ref7):
    """plans flagged LTP_STATUS_OVERFLOW (or whose rows would end beyond the tile) were never written, so the
    receding-horizon gather must carry their start state over and must not read outside the tile."""
    import torch
    D, lim, ltp, _ = ref7
    n = 400
    qm = ltp.generateQueries(n, seed=21)
    b = ltp.planSwitchTimesBatch(*qm)
    torch.cuda.synchronize()
    off = b.offsets.cpu().numpy().view(np.uint64)
    cap = int(off[150] - off[0]) + 8                     # deliberately small tile: plans >= 150 do not fit
    big = torch.full((int(off[-1]) + 64,), float("nan"), dtype=torch.float64, device="cuda")
    tile = big[:cap]
    ltp.sampleBatch(b, 0, n, tile)
    st = b.status.cpu().numpy()
    lens = b.traj_len.cpu().numpy()
    over = (st & amd.STATUS_OVERFLOW) != 0
    assert over[150:][lens[150:] > 0].all() and not over[:150].any()
    q1, v1, a1 = ltp.replanStates(b, 0, n, tile, 50)
    torch.cuda.synchronize()
    for p in range(n):
        if over[p] or lens[p] <= 0:
            assert torch.equal(q1[p], qm[1][p]) and torch.equal(v1[p], qm[2][p]) and torch.equal(a1[p], qm[3][p]), p
        else:
            q, v, a, _ = amd.unpack_trajectory(tile.cpu().numpy(), int(off[p] - off[0]), D, int(lens[p]))
            kk = min(50, int(lens[p]) - 1)
            assert np.array_equal(q1[p].cpu().numpy(), q[:, kk]) and np.array_equal(a1[p].cpu().numpy(), a[:, kk])
    assert not torch.isnan(q1).any()                     # nothing came from beyond the tile (NaN there)
    # the capacity test alone protects too: clear the flag, the gather still refuses rows beyond `capacity`
    b.status &= ~amd.STATUS_OVERFLOW
    q2, _, _ = ltp.replanStates(b, 0, n, tile, 50)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)


def test_one_handle_on_two_streams_serialises_on_its_workspace(amd, ref7):
    """the compaction queues / lane flags / scan scratch are per handle. Two streams that use ONE handle (a
    double-buffered pipeline) must get the results of serial runs: the library orders them with an event."""
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)       # reference limits: 6.5 % of lanes go through queue A
    streams = [torch.cuda.Stream() for _ in range(2)]
    n = 200000
    queries = [ltp.generateQueries(n, seed=300 + i) for i in range(2)]
    serial = []
    for i in range(2):
        b = ltp.planSwitchTimesBatch(*queries[i])
        torch.cuda.synchronize()
        serial.append({k: getattr(b, k).clone() for k in ("t_opt", "t_scaled", "v_drive", "mod", "traj_len", "status", "offsets")})
    batches = [None, None]
    for rep in range(6):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                batches[i] = ltp.planSwitchTimesBatch(*queries[i], batch=batches[i])
    torch.cuda.synchronize()
    for i in range(2):
        for k, want in serial[i].items():
            assert torch.equal(getattr(batches[i], k).nan_to_num(7.0) if want.is_floating_point() else getattr(batches[i], k),
                               want.nan_to_num(7.0) if want.is_floating_point() else want), (i, k)


def test_end_limit_verdict_without_sampling(amd, ref7, oracle_mod):
    """status == 0 must mean planTrajectory's `true` also when nothing is sampled (cc:59-61 needs the last sample)."""
    import torch
    D, lim, _, _ = ref7
    ltp4 = amd.LongTermPlanner(D, 0.004, device=0, **lim)      # 4 ms: ~0.5 % of random plans overshoot a joint limit
    orc4 = oracle_mod.Oracle(D, 0.004, **lim)
    n = 20000
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=77)
    o = orc4.plan_batch(qg, q0, v0, a0, sample=True)
    want_true = o["status"] == 1                               # oracle: 1 = true, 2 = end-limit false, 0 = early false
    assert (o["status"] == 2).sum() >= 3
    r0 = ltp4.planBatchHost(qg, q0, v0, a0, sample=False)
    assert np.array_equal(r0["status"] == 0, want_true)
    assert np.array_equal((r0["status"] & amd.STATUS_END_LIMIT) != 0, o["status"] == 2)
    # device path: plan alone carries the pre-sampling verdict, end_limit=True the full one, the sampler the same bits
    qm = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    b = ltp4.planSwitchTimesBatch(*qm)
    torch.cuda.synchronize()
    assert not (b.status.cpu().numpy() & amd.STATUS_END_LIMIT).any()
    b1 = ltp4.planSwitchTimesBatch(*qm, end_limit=True)
    torch.cuda.synchronize()
    st1 = b1.status.cpu().numpy().copy()
    tile = torch.zeros(int(b.offsets[-1].item()), dtype=torch.float64, device="cuda")
    ltp4.sampleBatch(b, 0, n, tile)
    torch.cuda.synchronize()
    assert np.array_equal(st1, b.status.cpu().numpy()) and np.array_equal(st1, r0["status"])
    # sub-range form
    b2 = ltp4.planSwitchTimesBatch(*qm)
    ltp4.endLimit(b2, 5000, 10000)
    torch.cuda.synchronize()
    st2 = b2.status.cpu().numpy()
    assert np.array_equal(st2[5000:15000], st1[5000:15000]) and not (st2[:5000] & amd.STATUS_END_LIMIT).any()


def test_changed_geometry_between_plan_and_sample_is_rejected(amd, ref7):
    """dof / t_sample / max_samples / sample_stride define a planned batch's records, offsets and row strides;
    changing one of them on the handle before a consumer call is an error, not overlapping rows."""
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qm = ltp.generateQueries(300, seed=2)
    b = ltp.planSwitchTimesBatch(*qm)
    tile = torch.zeros(int(b.offsets[-1].item()), dtype=torch.float64, device="cuda")
    for change, undo in ((lambda: ltp.setMaxSamples(64), lambda: ltp.setMaxSamples(0)),
                         (lambda: ltp.setSampleStride(2), lambda: ltp.setSampleStride(1)),
                         (lambda: ltp.setSampleTime(0.002), lambda: ltp.setSampleTime(0.001)),
                         (lambda: ltp.setDoF(D - 1), lambda: ltp.setDoF(D))):
        change()
        for call in (lambda: ltp.sampleBatch(b, 0, 300, tile), lambda: ltp.envelopeBatch(b, 0, 300, 32, 8),
                     lambda: ltp.replanStates(b, 0, 300, tile, 3), lambda: ltp.stateAt(b, 0, 300, 3), lambda: ltp.endLimit(b, 0, 300)):
            with pytest.raises(amd.LtpError) as e:
                call()
            assert e.value.code == 1 and "plan it again" in str(e.value)
        undo()
        ltp.sampleBatch(b, 0, 300, tile)                      # restored: accepted again
    torch.cuda.synchronize()


@pytest.mark.parametrize("limits,n", [("ref", 700), ("panda", 1500), ("ref30", 130)])
def test_table_pass_gives_the_bits_of_the_fused_build(amd, limits, n):
    """The sampler's run tables come either from the cooperative build inside k_sample or from the table pass
    (k_build_tables, lane = (plan, joint)): rows of every format, envelopes and the end-limit verdict must be bit-identical,
    also when the table workspace is so small that the range is processed in many pieces."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qm = ltp.generateQueries(n, seed=77)
    qm[1][11, 0] = 99.0                                         # a rejected plan in the middle
    # (1200, 1): most plans have all their runs inside the cap, i.e. more than the 8 runs the capped loader fetches up front
    for cap, stride in ((0, 1), (200, 1), (0, 3), (64, 2), (1200, 1)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        res = {}
        for mode in ("fused", "tables", "tables_small_workspace"):
            ltp.setTablePass(0, (1 << 32) if mode != "tables_small_workspace" else 40 * D * 912)
            b = ltp.planSwitchTimesBatch(*qm)
            total = int(b.offsets[-1].item())
            t64 = torch.full((total,), 3.0, dtype=torch.float64, device="cuda")
            t32 = torch.full((total,), 3.0, dtype=torch.float32, device="cuda")
            ltp.sampleBatch(b, 0, n, t64, tables=(mode != "fused"))
            ltp.sampleBatch(b, 0, n, t32, tables=(mode != "fused"), streaming=False)
            # a sub-range into its own tile: offsets are relative to the first plan of the call
            sub = torch.full((int((b.offsets[n - 3] - b.offsets[40]).item()) + 8,), 3.0, dtype=torch.float64, device="cuda")
            ltp.sampleBatch(b, 40, n - 43, sub, tables=(mode != "fused"))
            # block interleave factors that are not a power of two / larger than the range / 1: other item -> plan maps
            t48 = torch.full((total,), 3.0, dtype=torch.float64, device="cuda")
            ltp.sampleBatch(b, 0, n, t48, tables=(mode != "fused"), spread=48)
            t1 = torch.full((total,), 3.0, dtype=torch.float64, device="cuda")
            ltp.sampleBatch(b, 0, n, t1, tables=(mode != "fused"), spread=1 if mode == "tables" else 5000)
            torch.cuda.synchronize()
            assert torch.equal(t48, t64) and torch.equal(t1, t64), (cap, stride, mode, "spread")
            res[mode] = (t64, t32, sub, b.status.clone(), b.traj_len.clone())
        for mode in ("tables", "tables_small_workspace"):
            for got, want in zip(res[mode], res["fused"]):
                assert torch.equal(got, want), (cap, stride, mode)
    ltp.setMaxSamples(0); ltp.setSampleStride(1)
    b = ltp.planSwitchTimesBatch(*qm)
    env = {}
    for mode, flag in (("fused", -1), ("tables", 1), ("pieces", 1)):
        ltp.setTablePass(flag, (1 << 32) if mode != "pieces" else 40 * D * 912)
        env[mode] = (ltp.envelopeBatch(b, 0, n, 48, 24).clone(), ltp.envelopeBatch(b, 30, n - 60, 100, 7).clone())
    torch.cuda.synchronize()
    for mode in ("tables", "pieces"):
        for got, want in zip(env[mode], env["fused"]):
            assert torch.equal(got.nan_to_num(7.0), want.nan_to_num(7.0)), mode


@pytest.mark.parametrize("limits,ts", [("ref", 0.004), ("panda", 0.001), ("ref30", 0.002)])
def test_fused_small_batch_path_has_the_bits_of_the_batched_path(amd, limits, ts):
    """A call with n * dof <= 128 (a single planTrajectory above all) runs as ONE launch of one block writing into pinned
    host memory (k_plan_small). Records, offsets-relative rows, statuses and the one-joint getTrajectory entry must be
    bit-identical to what the batched kernels give for the same queries inside a large batch."""
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    n_big = 300
    qg, q0, v0, a0 = amd.generate_queries(n_big, lim, seed=5)
    q0[3, 0] = 99.0                                             # rejected by checkInputs
    qg[4] = q0[4]; v0[4] = 0.0; a0[4] = 0.0                     # the all-zero plan (traj_len 1)
    big = ltp.planBatchHost(qg, q0, v0, a0, sample=True)        # staged path: batched kernels
    small_n = max(1, min(128 // D, 9))
    for first in (0, 3, 50):
        for cnt in sorted({1, min(2, small_n), small_n}):
            sl = slice(first, first + cnt)
            r = ltp.planBatchHost(qg[sl], q0[sl], v0[sl], a0[sl], sample=True)
            r0 = ltp.planBatchHost(qg[sl], q0[sl], v0[sl], a0[sl], sample=False)
            for key in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status"):
                assert r[key].tobytes() == big[key][sl].tobytes(), (first, cnt, key)
                assert r0[key].tobytes() == big[key][sl].tobytes(), (first, cnt, key, "no rows")
            lo, hi = int(big["offsets"][first]), int(big["offsets"][first + cnt])
            assert np.array_equal(r["offsets"], big["offsets"][first:first + cnt + 1] - big["offsets"][first])
            assert r["packed"].tobytes() == big["packed"][lo:hi].tobytes(), (first, cnt)
            g = ltp.getTrajectoryBatchHost(big["t_scaled"][sl], big["dir"][sl], big["mod"][sl], q0[sl], v0[sl], a0[sl], big["v_drive"][sl])
            ran = (big["status"][sl] & 0x37) == 0
            assert np.array_equal(g["traj_len"][ran], big["traj_len"][sl][ran])
            if ran.all():
                assert g["packed"].tobytes() == big["packed"][lo:hi].tobytes()
    # capped and strided rows go through the same kernel
    ltp.setMaxSamples(100); ltp.setSampleStride(3)
    big2 = ltp.planBatchHost(qg, q0, v0, a0, sample=True)
    r = ltp.planBatchHost(qg[10:12], q0[10:12], v0[10:12], a0[10:12], sample=True)
    lo, hi = int(big2["offsets"][10]), int(big2["offsets"][12])
    assert r["packed"].tobytes() == big2["packed"][lo:hi].tobytes() and np.array_equal(r["status"], big2["status"][10:12])
    # a result larger than the pinned buffer (8 MiB) falls back to the staged path
    ltp.setMaxSamples(0); ltp.setSampleStride(1)
    if D == 30:
        slow = amd.LongTermPlanner(D, 0.0002, device=0, **lim)  # 5x the samples: > 1 Mi doubles per plan
        a = slow.planBatchHost(qg[:1], q0[:1], v0[:1], a0[:1], sample=True)
        b = slow.planBatchHost(qg[:40], q0[:40], v0[:40], a0[:40], sample=True)
        assert a["packed"].size * 8 > (8 << 20) and a["packed"].tobytes() == b["packed"][: int(b["offsets"][1])].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("limits,ts", [("ref", 0.004), ("panda", 0.001), ("ref30", 0.002)])
def test_host_threads_share_one_handle(amd, limits, ts):
    """Several host threads on ONE handle: single planTrajectory-sized calls (the one-launch path), the one-joint entry points,
    checkInputs and setLimits (unchanged limits) at the same time. The library serialises them (lock order host_mu before mu
    everywhere, ltp_capi: ltp_set_limits); every caller gets its own unchanged result and nothing deadlocks (the round-3 advisor
    found an ABBA pair here; this test fails by timeout if one comes back)."""
    import threading
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    qg, q0, v0, a0 = amd.generate_queries(60, lim, seed=11)
    q0[3, 0] = 99.0                                             # rejected by checkInputs
    qg[4] = q0[4]; v0[4] = 0.0; a0[4] = 0.0                     # the all-zero plan
    small_n = max(1, min(128 // D, 9))
    cases = [(first, cnt, rows) for first in (0, 3, 4, 17) for cnt in sorted({1, small_n}) for rows in (True, False)]
    want = [ltp.planBatchHost(qg[f:f + c], q0[f:f + c], v0[f:f + c], a0[f:f + c], sample=r) for f, c, r in cases]
    keys = ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status", "offsets")
    brake = ltp.optBraking(0, 0.4 * lim["v_max"][0], -0.5 * lim["a_max"][0])
    scale = ltp.timeScaling(0, 0.6, 0.1, 0.2, 0.3, 1.0, 2.5)
    errors = []

    def plans(tid):
        try:
            for rep in range(25):
                f, c, r = cases[(tid * 7 + rep) % len(cases)]
                w = want[(tid * 7 + rep) % len(cases)]
                g = ltp.planBatchHost(qg[f:f + c], q0[f:f + c], v0[f:f + c], a0[f:f + c], sample=r)
                for key in keys:
                    assert g[key].tobytes() == w[key].tobytes(), (tid, rep, key)
                if r:
                    assert g["packed"].tobytes() == w["packed"].tobytes(), (tid, rep)
        except Exception as e:                                   # noqa: BLE001 - reported below
            errors.append(repr(e))

    def lanes():
        try:
            for rep in range(40):
                b = ltp.optBraking(0, 0.4 * lim["v_max"][0], -0.5 * lim["a_max"][0])
                assert b[1] == brake[1] and np.array_equal(b[2], brake[2]) and b[3] == brake[3]
                sc = ltp.timeScaling(0, 0.6, 0.1, 0.2, 0.3, 1.0, 2.5)
                assert np.array_equal(np.asarray(sc[1]), np.asarray(scale[1]), equal_nan=True)
                assert ltp.checkInputs(q0[0], v0[0], a0[0])
        except Exception as e:                                   # noqa: BLE001
            errors.append(repr(e))

    def setters():
        try:
            for rep in range(20):
                ltp.setLimits(**lim)
        except Exception as e:                                   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=plans, args=(0,), daemon=True), threading.Thread(target=plans, args=(1,), daemon=True),
               threading.Thread(target=lanes, daemon=True), threading.Thread(target=setters, daemon=True)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "host threads on one handle did not finish: deadlock"
    assert not errors, errors[:3]
    # the one-joint getTrajectory entry takes the one-launch path too
    f, c = 17, 1
    w = want[cases.index((17, 1, True))]
    g = ltp.getTrajectoryBatchHost(w["t_scaled"], w["dir"], w["mod"], q0[f:f + c], v0[f:f + c], a0[f:f + c], w["v_drive"])
    if int(w["status"][0]) & 0x37 == 0:
        assert g["packed"].tobytes() == w["packed"].tobytes()
    # new limits change the result of the next call
    half = dict(lim); half["v_max"] = [0.5 * v for v in lim["v_max"]]
    ltp.setLimits(**half)
    a = ltp.planBatchHost(qg[:1], q0[:1], v0[:1], a0[:1], sample=True)
    assert a["t_scaled"].tobytes() != want[0]["t_scaled"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("limits,n,cap", [("panda", 70001, 64), ("ref30", 66000, 16)])
def test_table_pass_matches_the_fused_sampler_on_a_large_range(amd, limits, n, cap):
    """The short-row sampler on a range of tens of thousands of plans (every resident block draws many items, the table
    workspace is reused by calls right behind each other and from a second stream of the same handle): rows and statuses must
    equal the fused sampler's."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qm = ltp.generateQueries(n, seed=5)
    qm[1][4242, 0] = 99.0                                        # a rejected plan
    ltp.setMaxSamples(cap)
    res = {}
    for mode in ("fused", "tables"):
        b = ltp.planSwitchTimesBatch(*qm)
        total = int(b.offsets[-1].item())
        t64 = torch.full((total,), 3.0, dtype=torch.float64, device="cuda")
        t32 = torch.full((total,), 3.0, dtype=torch.float32, device="cuda")
        ltp.sampleBatch(b, 0, n, t64, tables=(mode != "fused"))
        ltp.sampleBatch(b, 0, n, t32, tables=(mode != "fused"))          # right behind: the table workspace is reused
        s2 = torch.cuda.Stream()
        sub = torch.full((int((b.offsets[n - 5] - b.offsets[100]).item()),), 3.0, dtype=torch.float64, device="cuda")
        with torch.cuda.stream(s2):                                       # another stream of the same handle: serialised by the library
            s2.wait_stream(torch.cuda.current_stream())
            ltp.sampleBatch(b, 100, n - 105, sub, tables=(mode != "fused"))
        torch.cuda.synchronize()
        res[mode] = (t64, t32, sub, b.status.clone())
    for got, want in zip(res["tables"], res["fused"]):
        assert torch.equal(got, want)


@pytest.mark.parametrize("limits,dof,n", [("panda", None, 3001), ("ref", 3, 2000), ("ref", 6, 1001), ("ref", 1, 700)])
def test_short_rows_table_pass_and_walk_kernel_give_the_bits_of_the_fused_sampler(amd, limits, dof, n):
    """Rows of a few dozen samples through all three samplers: the fused one, the table pass (k_build_tables + k_sample_tab_*, whole
    and in pieces of a small workspace) and the walk kernel (compact batches; plans with more than 8 runs inside the cap — short
    moves from rest — rebuilt as wide batches). Whatever the cap, stride, range, odd counts, rejected neighbours or tiles too small
    for some plans: rows, statuses and lengths agree bit for bit. (Round 4 also had a two-plans-per-item form of the table sampler,
    profiles/EXPERIMENTS.md E6.3; the walk kernel superseded it and it was removed.)"""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=5))
    q0[11, 0] = 99.0                                            # rejected plans, alone and as neighbours
    q0[14, 0] = 99.0
    q0[20, 0] = 99.0; q0[21, 0] = 99.0
    # short moves from rest: trajectories of a few dozen samples whose runs ALL lie inside the cap (more than 8 per joint)
    short = torch.arange(100, min(n, 900), 7, device=qg.device)
    qg[short] = q0[short] + 0.02 * torch.sign(qg[short] - q0[short] + 1e-9)    # (beyond the 4e-3 rad early exit of cc:104-109)
    v0[short] = 0.0
    a0[short] = 0.0
    lo = torch.tensor(lim["q_min"], dtype=torch.float64, device=qg.device)
    hi = torch.tensor(lim["q_max"], dtype=torch.float64, device=qg.device)
    qg[short] = torch.minimum(torch.maximum(qg[short], lo), hi)
    for cap, stride, f32 in ((64, 1, False), (63, 1, False), (1, 1, False), (2, 1, False), (17, 3, False), (33, 1, False), (128, 1, True), (5, 2, True), (100, 1, True)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        total = int(b.offsets[-1].item())
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("fused", "tables", "tables_pieces", "walk"):
            ltp.setTablePass(0, (1 << 32) if mode != "tables_pieces" else 37 * D * 912)
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            full = torch.full((total,), 3.0, dtype=dt, device="cuda")
            kw = dict(walk=True) if mode == "walk" else dict(tables=(mode != "fused"), walk=False)
            ltp.sampleBatch(b, 0, n, full, **kw)
            kern = ltp.lastSamplerKernel()
            assert ("tab" in kern) == mode.startswith("tables") and ("walk" in kern) == (mode == "walk"), (mode, kern)
            # an odd sub-range into its own tile; and a tile too small for the last plans
            sub = torch.full((int((b.offsets[n - 2] - b.offsets[41]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 41, n - 43, sub, spread=48, **kw)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2].item()) + 5,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=False, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for mode in ("tables", "tables_pieces", "walk"):
            for k, (got, want) in enumerate(zip(res[mode], res["fused"])):
                if k == 5 and mode != "walk":        # (k_sample_walk never walks a plan that does not fit: the fused sampler's statuses)
                    # a plan that does not fit the tile: the table pass has walked it to its end (END_LIMIT set), the fused sampler
                    # never built its tables; everything else about the statuses is equal, and the table samplers agree exactly
                    skipped = (want & 32) != 0
                    assert torch.equal(got & ~torch.where(skipped, 8, 0), want), (cap, stride, f32, mode, k)
                    assert torch.equal(got, res["tables"][5]), (cap, stride, f32, mode, "statuses of the table samplers")
                    assert skipped.any()
                    continue
                assert torch.equal(got, want), (cap, stride, f32, mode, k)
    ltp.setTablePass(0, 1 << 32)


@pytest.mark.parametrize("limits,dof,n", [("panda", None, 2503), ("ref", 28, 401), ("ref", 2, 1500), ("ref", 9, 900), ("ref", 30, 301),
                                           ("ref", 63, 140)])
def test_walk_sampler_keeps_its_tables_in_the_compute_unit_and_gives_the_same_rows(amd, limits, dof, n):
    """k_sample_walk_*: capped rows (<= 63 joints) with the run tables built by a builder wave inside the
    sampler's block — no table pass, no table traffic. It is what the library takes automatically for such rows; rows, statuses and
    lengths must be those of the fused sampler bit for bit: every cap / stride / element type, batches of 9 / 2 / 9 / 7 / 2 / 1 plans (wide: 4 / 1 / 14 / 3 plans, then 28 + 2 joints and
    28 + 28 + 7 joints of one plan at a time),
    ranges that start anywhere, tiles too small for the last plans, rejected plans, and plans with more than 8 runs inside the cap
    (the list pass through the fused kernel)."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=9))
    q0[7, 0] = 99.0; q0[8, 0] = 99.0; q0[30, D - 1] = 99.0
    short = torch.arange(60, min(n, 800), 5, device=qg.device)
    qg[short] = q0[short] + 0.02 * torch.sign(qg[short] - q0[short] + 1e-9)
    v0[short] = 0.0
    a0[short] = 0.0
    qg[short] = torch.minimum(torch.maximum(qg[short], torch.tensor(lim["q_min"], dtype=torch.float64, device=qg.device)),
                              torch.tensor(lim["q_max"], dtype=torch.float64, device=qg.device))
    for cap, stride, f32 in ((256, 1, False), (255, 1, True), (129, 2, False), (64, 1, False), (200, 4, True), (1, 1, False), (31, 1, True)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("fused", "walk", "auto"):
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            total = int(b.offsets[-1].item())
            kw = dict(tables=False, walk=False) if mode == "fused" else (dict(walk=True) if mode == "walk" else {})
            full = torch.full((total,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **kw)
            kern = ltp.lastSamplerKernel()
            assert ("walk" in kern) == (mode != "fused"), (mode, kern, cap)          # automatic choice: the walk kernel for these caps
            sub = torch.full((int((b.offsets[n - 2] - b.offsets[13]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 13, n - 15, sub, spread=48, **kw)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2 + 3].item()) + 5,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=False, spread=1, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for mode in ("walk", "auto"):
            for k, (got, want) in enumerate(zip(res[mode], res["fused"])):
                assert torch.equal(got, want), (cap, stride, f32, mode, k)
        assert (res["fused"][5] & 32).any(), "the small tile did not leave any plan out"
    # what the library takes by itself (want_walk): caps up to 768 samples, float32 rows, every 3rd sample or sparser -> the walk
    # kernel; whole or long float64 rows at stride 1-2 -> the fused sampler. Either can be forced; the rows are the same.
    for cap, stride, f32, walk_by_default in ((257, 1, False, True), (768, 2, False, True), (769, 1, False, False), (1025, 1, False, False), (1025, 1, True, True),
                                              (0, 1, False, False), (0, 2, False, False), (0, 3, False, True), (0, 1, True, True), (3000, 5, False, True)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        dt = torch.float32 if f32 else torch.float64
        t0 = torch.full((int(b.offsets[-1].item()),), 3.0, dtype=dt, device="cuda")
        ltp.sampleBatch(b, 0, n, t0)
        assert ("walk" in ltp.lastSamplerKernel()) == walk_by_default, (cap, stride, f32, ltp.lastSamplerKernel())
        t1 = torch.full_like(t0, 3.0)
        ltp.sampleBatch(b, 0, n, t1, **(dict(tables=False, walk=False) if walk_by_default else dict(walk=True)))
        assert ("walk" in ltp.lastSamplerKernel()) == (not walk_by_default), (cap, stride, f32, ltp.lastSamplerKernel())
        assert torch.equal(t0, t1), (cap, stride, f32)
        del t0, t1


def test_walk_sampler_random_configurations_against_the_fused_sampler(amd):
    """A randomized differential test of k_sample_walk_* (the default writer of capped rows) against the fused sampler: 48 random
    combinations of joints (1-63), sample time, cap (1-2500), stride (1-6), element type, batch size, sub-range, block interleave and
    tile size, on random limit sets (fast and slow jerk, i.e. many and few runs inside the cap) with rejected plans and short moves
    mixed in; every 7th configuration in MATLAB semantics (against the table pass). Rows, statuses and lengths must agree bit for bit.
    (Round 5: 2 %, 30 % or 70 % of the plans rejected and dead stretches of several queue items — gathered batches — and caps either
    side of the autonomous form's and the gathered items' limits.)"""
    import torch
    # (a soak: LTP_WALK_TRIALS=3000 LTP_WALK_SEED=1 python -m pytest tests/test_gpu_edge.py -k random_configurations)
    rng = np.random.default_rng(20260401 + int(os.environ.get("LTP_WALK_SEED", "0")))
    for trial in range(int(os.environ.get("LTP_WALK_TRIALS", "48"))):
        D = int(rng.integers(1, 29)) if trial % 4 else int(rng.integers(29, 64))
        ts = float(rng.choice([0.0005, 0.001, 0.002, 0.004]))
        v_max = rng.uniform(0.5, 3.0, D)
        a_max = rng.uniform(1.0, 20.0, D)
        j_max = a_max * (rng.uniform(5.0, 600.0, D) if trial % 3 else rng.uniform(0.5, 5.0, D))
        q_hi = rng.uniform(1.0, 3.5, D)
        lim = dict(q_min=list(-q_hi), q_max=list(q_hi), v_max=list(v_max), a_max=list(a_max), j_max=list(j_max))
        ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
        matlab = trial % 7 == 3                                  # MATLAB semantics: no fused sampler there, the partner is the table pass
        if matlab:
            ltp.setSemantics("matlab")
        n = int(rng.integers(1, 900))
        qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=1000 + trial))
        # rejected plans: a few, a third or most of them (round 5: the walk kernels build their batches from an item's live plans)
        for p in rng.integers(0, n, size=max(1, int(n * float(rng.choice([0.02, 0.02, 0.3, 0.7]))))):
            q0[int(p), 0] = 99.0
        if trial % 5 == 2 and n > 120:
            q0[40:120, 0] = 99.0                                 # a dead stretch of several queue items
        short = torch.as_tensor(rng.integers(0, n, size=max(1, n // 6)), device=qg.device)
        qg[short] = torch.clamp(q0[short] + float(rng.choice([0.01, 0.03, 0.1])) * torch.sign(qg[short] - q0[short] + 1e-9), -99.0, 99.0)
        qg[short] = torch.minimum(torch.maximum(qg[short], torch.tensor(lim["q_min"], dtype=torch.float64, device=qg.device)),
                                  torch.tensor(lim["q_max"], dtype=torch.float64, device=qg.device))
        v0[short] = 0.0
        a0[short] = 0.0
        cap = int(rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 255, 256, 500, 1024, 1025, 2500]))
        stride = int(rng.integers(1, 7))
        f32 = bool(rng.integers(0, 2))
        spread = int(rng.choice([0, 1, 3, 48, 5000]))
        first = int(rng.integers(0, n))
        count = int(rng.integers(0, n - first + 1))
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("fused", "walk"):
            kw = (dict(walk=False) if matlab else dict(tables=False, walk=False)) if mode == "fused" else dict(walk=True)
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            need = int((b.offsets[first + count] - b.offsets[first]).item())
            tile = torch.full((need + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, first, count, tile, spread=spread, **kw)
            if count:
                assert ("walk" in ltp.lastSamplerKernel()) == (mode == "walk"), (trial, ltp.lastSamplerKernel())
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2].item()) + 3,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=bool(trial & 1), **kw)
            torch.cuda.synchronize()
            res[mode] = (tile, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for k, (got, want) in enumerate(zip(res["walk"], res["fused"])):
            assert torch.equal(got, want), (trial, D, ts, cap, stride, f32, n, first, count, spread, matlab, k)
        if trial % 3 == 1:
            # the same rows without the end-limit verdict (flags bit 4): every status bit but END_LIMIT as before
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            st0 = b.status.clone()
            need = int((b.offsets[first + count] - b.offsets[first]).item())
            tile = torch.full((need + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, first, count, tile, spread=spread, walk=True, verdict=False)
            torch.cuda.synchronize()
            assert torch.equal(tile, res["walk"][0]), (trial, "no-verdict rows")
            keep = ~amd.STATUS_END_LIMIT
            full_st = res["walk"][2]
            sel = torch.zeros(n, dtype=torch.bool, device=st0.device)
            sel[first:first + count] = True
            assert torch.equal(b.status[sel] & keep, full_st[sel] & keep) and torch.equal(b.status[~sel], st0[~sel]), (trial, "no-verdict status")


@pytest.mark.parametrize("limits,dof,n", [("panda", None, 1201), ("ref", None, 400), ("ref", 30, 60), ("ref", 3, 500)])
def test_walk_sampler_whole_rows_equal_the_fused_sampler(amd, limits, dof, n):
    """k_sample_walk_* on request (flag bit 6) for rows WITHOUT a cap: wide batches only, one row per wave pass, windowed row
    descriptors. Every stride / element type, a range that starts anywhere, a tile too small for the last plans, rejected plans:
    rows, statuses and lengths are those of the fused sampler bit for bit."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=19))
    q0[5, 0] = 99.0; q0[6, D - 1] = 99.0; q0[n - 1, 0] = 99.0
    ltp.setMaxSamples(0)
    for stride, f32 in ((1, False), (2, False), (4, False), (1, True), (3, True), (7, False)):
        ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("fused", "walk"):
            kw = dict(tables=False, walk=False) if mode == "fused" else dict(walk=True)
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            full = torch.full((int(b.offsets[-1].item()),), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **kw)
            assert ("walk" in ltp.lastSamplerKernel()) == (mode == "walk"), (mode, ltp.lastSamplerKernel())
            sub = torch.full((int((b.offsets[n - 3] - b.offsets[11]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 11, n - 14, sub, spread=7, **kw)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2 + 1].item()) + 5,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=False, spread=1, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for k, (got, want) in enumerate(zip(res["walk"], res["fused"])):
            assert torch.equal(got, want), (stride, f32, k)
        assert (res["fused"][5] & 32).any(), "the small tile did not leave any plan out"
        del res


def test_rows_longer_than_a_descriptor_window(amd):
    """A row of 160 M samples (Ts = 40 ns): longer than the 2^27 samples one buffer descriptor of the fused sampler spans and than the
    2^25 of the walk kernel's long-row form — both advance a window along the row. The two samplers must agree bit for bit, and the
    samples either side of every window boundary and the last one must be the states k_state_at computes from the records alone."""
    import torch
    D, lim = amd.limit_set("ref", 2)
    ltp = amd.LongTermPlanner(D, 4e-8, device=0, **lim)
    qg = torch.tensor([[3.0, -2.5], [0.3, 0.1]], dtype=torch.float64, device="cuda")
    q0 = torch.tensor([[-3.0, 2.5], [0.2, 0.0]], dtype=torch.float64, device="cuda")
    v0 = torch.zeros_like(q0)
    a0 = torch.zeros_like(q0)
    b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
    L = int(b.traj_len[0].item())
    assert L > (1 << 27) + 1000, L
    total = int(b.offsets[-1].item())
    tiles = {}
    for mode in ("fused", "walk"):
        bb = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        t = torch.full((total,), 3.0, dtype=torch.float64, device="cuda")
        ltp.sampleBatch(bb, 0, 2, t, **(dict(walk=True) if mode == "walk" else dict(tables=False, walk=False)))
        assert ("walk" in ltp.lastSamplerKernel()) == (mode == "walk")
        torch.cuda.synchronize()
        assert int(bb.status[0].item()) & ~8 == 0
        tiles[mode] = t
    assert torch.equal(tiles["fused"], tiles["walk"])
    stride = (L + 31) // 32 * 32
    rows = tiles["walk"][: 4 * D * stride].view(4, D, stride)
    for k in ((1 << 25) - 1, 1 << 25, (1 << 25) + 1, (1 << 26) - 1, 1 << 26, (1 << 27) - 1, 1 << 27, (1 << 27) + 1, L - 2, L - 1):
        q, v, a = ltp.stateAt(b, 0, 1, k)
        for x, got in enumerate((q, v, a)):
            assert torch.equal(got[0], rows[x, :, k]), (k, x)
    del tiles


@pytest.mark.parametrize("dof", [64, 100])
def test_walk_sampler_beyond_63_joints(amd, dof):
    """More joints than a compact batch has lanes: an item is one plan, taken 63 joints at a time (wide: 28). Capped, sparse, float32
    and whole rows against the fused sampler, bit for bit."""
    import torch
    half = dof // 2
    D, lim = amd.limit_set("ref", dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    Dh, limh = amd.limit_set("ref", half)
    gen = amd.LongTermPlanner(Dh, 0.001, device=0, **limh)
    n = 70
    parts = [gen.generateQueries(n, seed=s) for s in (3, 4)]
    qg, q0, v0, a0 = (torch.cat([parts[0][k], parts[1][k]], dim=1).contiguous() for k in range(4))
    q0[4, dof - 1] = 99.0
    short = torch.arange(10, n, 6, device=qg.device)
    qg[short] = torch.clamp(q0[short] + 0.02, -3.0, 3.0)
    v0[short] = 0.0
    a0[short] = 0.0
    for cap, stride, f32 in ((128, 1, False), (16, 1, True), (0, 4, False), (0, 1, True), (600, 2, False), (0, 1, False)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode in ("fused", "walk"):
            kw = dict(tables=False, walk=False) if mode == "fused" else dict(walk=True)
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            full = torch.full((int(b.offsets[-1].item()),), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **kw)
            assert ("walk" in ltp.lastSamplerKernel()) == (mode == "walk"), (mode, ltp.lastSamplerKernel())
            sub = torch.full((int((b.offsets[n - 2] - b.offsets[5]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 5, n - 9, sub, spread=3, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, b.status.clone(), b.traj_len.clone())
        for k, (got, want) in enumerate(zip(res["walk"], res["fused"])):
            assert torch.equal(got, want), (cap, stride, f32, k)
        assert int((res["walk"][3] > 0).sum().item()) >= n - 1



@pytest.mark.parametrize("limits,dof,n,semantics", [("panda", None, 4000, "cpp"), ("ref", None, 3000, "cpp"), ("ref", 30, 400, "cpp"), ("panda", None, 2000, "matlab")])
def test_analytic_envelopes_agree_with_the_exhaustive_form(amd, oracle_mod, limits, dof, n, semantics):
    """ltp_set_envelope_mode(LTP_ENVELOPE_ANALYTIC) evaluates, per run and window, the samples at the ends of the
    stretch and either side of the real roots of q'(m) instead of every sample. Those candidates are samples of the row, so the result
    can differ from the exhaustive (bit-exact) form only where a neighbour undercuts by rounding alone: <= 1e-12 here (and equal in
    nearly every window), NaN envelopes and statuses identical; C++ semantics also within 1e-9 of the oracle's reduced rows. Windows of
    1 .. 700 samples, through the table pass and the fused build, short moves, rejected plans, windows past the end."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ltp.setSemantics(semantics)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=88))
    v0[5, 0] = 99.0                                              # a rejected plan: NaN envelope
    short = torch.arange(30, min(n, 900), 7, device=qg.device)
    qg[short] = q0[short] + 0.01 * torch.sign(qg[short] - q0[short] + 1e-9)
    v0[short] = 0.0
    a0[short] = 0.0
    qg[short] = torch.minimum(torch.maximum(qg[short], torch.tensor(lim["q_min"], dtype=torch.float64, device=qg.device)),
                              torch.tensor(lim["q_max"], dtype=torch.float64, device=qg.device))
    orc = oracle_mod.Oracle(D, 0.001, **lim) if semantics == "cpp" else None
    worst, same, total, worst_oracle = 0.0, 0, 0, 0.0
    # table_pass 0: the analytic form is the register walk k_envelope_walk (no tables); 1 / -1: the block-cooperative k_envelope's analytic
    # form through the table pass / with the build inside the kernel
    for window, n_windows, table_pass in ((64, 32, 0), (1, 40, 0), (2, 64, 0), (3, 50, -1), (700, 4, 0), (129, 20, -1), (4000, 2, 0), (64, 32, 1), (5, 300, 0), (37, 9, 1)):
        if semantics == "matlab" and table_pass < 0:
            continue                                             # MATLAB semantics: envelopes always take the table pass
        ltp.setTablePass(table_pass)
        res = {}
        for mode in ("exhaustive", "analytic"):
            ltp.setEnvelopeMode(mode)
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            env = ltp.envelopeBatch(b, 3, n - 7, window, n_windows)
            assert ("analytic" in ltp.lastSamplerKernel()) == (mode == "analytic"), ltp.lastSamplerKernel()
            assert ("k_envelope_walk" in ltp.lastSamplerKernel()) == (mode == "analytic" and table_pass == 0), ltp.lastSamplerKernel()
            torch.cuda.synchronize()
            res[mode] = (env.cpu().numpy(), b.status.cpu().numpy(), b.traj_len.cpu().numpy())
        ex, an = res["exhaustive"][0], res["analytic"][0]
        assert np.array_equal(np.isnan(ex), np.isnan(an)) and np.array_equal(res["exhaustive"][1], res["analytic"][1])
        assert np.isnan(ex[5 - 3]).all()
        ok = ~np.isnan(ex)
        worst = max(worst, float(np.max(np.abs(ex[ok] - an[ok]))))
        same += int((ex[ok] == an[ok]).sum()); total += int(ok.sum())
        assert np.all(an[..., 0][ok[..., 0]] >= ex[..., 0][ok[..., 0]]) and np.all(an[..., 1][ok[..., 1]] <= ex[..., 1][ok[..., 1]])   # a subset of the samples
        if orc is not None and window in (64, 129):
            lens = res["analytic"][2]
            qgh, q0h, v0h, a0h = (x.cpu().numpy() for x in (qg, q0, v0, a0))
            for p in range(3, n - 4, 97):
                if lens[p] == 0:
                    continue
                o = orc.plan_trajectory(qgh[p], q0h[p], v0h[p], a0h[p])
                qo, L = o["q"], o["length"]
                pad = np.concatenate([qo, np.repeat(qo[:, -1:], max(window * n_windows - L, 0) + window, axis=1)], axis=1)
                wo = pad[:, : window * n_windows].reshape(D, n_windows, window)
                worst_oracle = max(worst_oracle, float(np.max(np.abs(an[p - 3, :, :, 0] - wo.min(axis=2)))), float(np.max(np.abs(an[p - 3, :, :, 1] - wo.max(axis=2)))))
    print(f"analytic vs exhaustive envelopes: worst |d| {worst:.2e}, identical in {same} of {total} values; vs the oracle's reduced rows {worst_oracle:.2e}")
    assert worst <= 1e-12 and same >= 0.999 * total and worst_oracle <= TOL


@pytest.mark.parametrize("limits,n", [("panda", 1200), ("ref", 500), ("ref30", 60)])
def test_walk_and_table_samplers_against_the_oracle_directly(amd, oracle_mod, limits, n):
    """k_sample_walk_* and the table-pass sampler were proven by a chain (== the fused sampler bit for bit, which is
    compared with the oracle). Here they meet the oracle's getTrajectory (cc:706-841) themselves: whole float64 rows, every 4th sample,
    float32 rows (the oracle's value rounded once, give or take the float32 ulp a 1e-12 difference can flip) and capped rows, each through
    the walk kernel and through the table pass, every sample of every plan, bar 1e-9."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    orc = oracle_mod.Oracle(D, 0.001, **lim)
    qg, q0, v0, a0 = amd.generate_queries(n, lim, seed=1234)
    dev = [torch.from_numpy(x).cuda() for x in (qg, q0, v0, a0)]
    o = orc.plan_batch(qg, q0, v0, a0, sample=False)
    ref = {}
    for p in range(n):
        if o["status"][p] != 0:
            ref[p] = orc.get_trajectory(o["t_scaled"][p], o["dir"][p], o["mod"][p], q0[p], v0[p], a0[p], o["v_drive"][p])
    worst = {}
    for cap, stride, f32 in ((0, 1, False), (0, 4, False), (0, 1, True), (100, 1, False), (40, 3, True)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        for mode, kw in (("walk", dict(walk=True)), ("tables", dict(tables=True, walk=False))):
            b = ltp.planSwitchTimesBatch(*dev)
            tile = torch.zeros(int(b.offsets[-1].item()), dtype=torch.float32 if f32 else torch.float64, device="cuda")
            ltp.sampleBatch(b, 0, n, tile, **kw)
            kern = ltp.lastSamplerKernel()
            assert ("walk" in kern) if mode == "walk" else ("tab" in kern), (mode, kern)
            torch.cuda.synchronize()
            off = b.offsets.cpu().numpy().view(np.uint64)
            lens = b.traj_len.cpu().numpy()
            host = tile.cpu().numpy()
            w = 0.0
            for p, (L, q, v, a, j) in ref.items():
                assert lens[p] == L
                stored = -(-L // stride)
                stored = min(stored, cap) if cap else stored
                got = amd.unpack_trajectory(host, int(off[p]), D, stored)
                for g, r in zip(got, (q, v, a, j)):
                    want = r[:, ::stride][:, :stored]
                    d = np.abs(g.astype(np.float64) - want)
                    if f32:
                        d = np.maximum(d - np.spacing(np.abs(want).astype(np.float32)).astype(np.float64), 0.0)   # one float32 ulp of rounding
                    w = max(w, float(d.max()))
            worst[(cap, stride, f32, mode)] = w
    print("walk / table samplers vs the oracle:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) <= TOL


@pytest.mark.parametrize("limits,dof,n,semantics", [("panda", None, 3000, "cpp"), ("ref", 30, 120, "cpp"), ("ref", 9, 500, "matlab")])
def test_autonomous_waves_write_the_rows_of_the_builder_form(amd, limits, dof, n, semantics):
    """Round 5: for caps of at most 32 samples the walk kernel runs as k_sample_walk_auto_* — every wave builds AND writes its own
    batches (eight walks per compute unit instead of four). Same walk_build / walk_stream functions: rows, statuses and lengths are
    those of the builder / streaming-wave form (flag bit 7) and of the fused sampler / table pass, bit for bit — caps 1-33, strides,
    both element types, sub-ranges, small tiles, rejected plans, trajectories that end inside the cap (wide batches)."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ltp.setSemantics(semantics)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=51))
    v0[7, 0] = 99.0; v0[8, D - 1] = 99.0
    short = torch.arange(20, min(n, 700), 5, device=qg.device)
    qg[short] = q0[short] + 0.002 * torch.sign(qg[short] - q0[short] + 1e-9)
    v0[short] = 0.0
    a0[short] = 0.0
    qg[short] = torch.minimum(torch.maximum(qg[short], torch.tensor(lim["q_min"], dtype=torch.float64, device=qg.device)),
                              torch.tensor(lim["q_max"], dtype=torch.float64, device=qg.device))
    other = dict(walk=False) if semantics == "matlab" else dict(tables=False, walk=False)
    for cap, stride, f32 in ((16, 1, False), (1, 1, False), (4, 1, True), (15, 2, False), (16, 3, True), (17, 1, False), (32, 1, False), (31, 2, True), (33, 1, False),
                             (4, 1 << 27, False)):      # cap x stride beyond 2^28: a compact slot's start has 28 bits -> wide batches, builder form
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode, kw in (("other", other), ("builder", dict(walk=True, auto_waves=False)), ("auto", {})):
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            full = torch.full((int(b.offsets[-1].item()),), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **kw)
            kern = ltp.lastSamplerKernel()
            assert ("auto" in kern) == (mode == "auto" and cap <= 32 and cap * stride < (1 << 28)), (mode, kern, cap)
            sub = torch.full((int((b.offsets[n - 2] - b.offsets[13]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 13, n - 15, sub, spread=11, **kw)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2 + 3].item()) + 5,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=False, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for mode in ("builder", "auto"):
            for k, (got, want) in enumerate(zip(res[mode], res["other"])):
                assert torch.equal(got, want), (cap, stride, f32, mode, k)
        assert (res["other"][5] & 32).any(), "the small tile did not leave any plan out"


@pytest.mark.gpu
@pytest.mark.parametrize("limits,dof,n,semantics", [("panda", None, 5003, "cpp"), ("ref", 30, 333, "cpp"), ("ref", 2, 4000, "cpp"), ("ref", 64, 400, "cpp"),
                                                     ("panda", None, 2500, "matlab")])
def test_walk_sampler_gathers_the_live_plans_of_an_item(amd, limits, dof, n, semantics):
    """Round 5: for caps of at most 64 samples a queue item of the walk kernels is three batches' worth of consecutive plans and a
    batch is made of the item's LIVE plans only (rejected plans have no rows and take no lane of a walk). With a third of the plans
    dead at random, long dead stretches (whole items, item tails and heads) and single survivors, rows, statuses and lengths are
    those of the fused sampler / table pass bit for bit — both kernel forms, sub-ranges that start and end inside dead stretches,
    small interleaves, a tile too small for the last plans."""
    import torch
    D, lim = amd.limit_set(limits, dof)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    ltp.setSemantics(semantics)
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=77))
    g = torch.Generator(device="cpu").manual_seed(5)
    dead = torch.rand(n, generator=g) < 0.33
    dead[100:190] = True                        # several whole items
    dead[200:230] = True; dead[214] = False     # a single survivor in a dead stretch
    dead[n - 40:] = True                        # the call ends in dead plans
    dead[:3] = True                             # and starts with some
    dead[300:330] = False                       # a stretch without any
    dead = dead.to(qg.device)
    v0[dead, 0] = 99.0                          # checkInputs rejects the plan (cc:63-80): traj_len 0, no rows
    other = dict(walk=False) if semantics == "matlab" else dict(tables=False, walk=False)
    for cap, stride, f32 in ((64, 1, False), (32, 1, False), (16, 1, False), (5, 1, True), (20, 3, False), (65, 1, False)):
        ltp.setMaxSamples(cap); ltp.setSampleStride(stride)
        dt = torch.float32 if f32 else torch.float64
        res = {}
        for mode, kw in (("other", other), ("builder", dict(walk=True, auto_waves=False)), ("default walk", dict(walk=True))):
            b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            assert int((b.traj_len == 0).sum().item()) >= int(dead.sum().item())
            full = torch.full((int(b.offsets[-1].item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, 0, n, full, **kw)
            assert ("walk" in ltp.lastSamplerKernel()) == (mode != "other")
            lo, hi = 105, n - 20                # starts and ends inside dead stretches
            sub = torch.full((int((b.offsets[hi] - b.offsets[lo]).item()) + 8,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b, lo, hi - lo, sub, spread=7, **kw)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            small = torch.full((int(b2.offsets[n // 2 + 3].item()) + 5,), 3.0, dtype=dt, device="cuda")
            ltp.sampleBatch(b2, 0, n, small, streaming=False, **kw)
            torch.cuda.synchronize()
            res[mode] = (full, sub, small, b.status.clone(), b.traj_len.clone(), b2.status.clone())
        for mode in ("builder", "default walk"):
            for k, (got, want) in enumerate(zip(res[mode], res["other"])):
                assert torch.equal(got, want), (cap, stride, f32, mode, k)
        assert (res["other"][5] & 32).any(), "the small tile did not leave any plan out"


@pytest.mark.gpu
def test_capped_rows_without_the_end_limit_verdict(amd, ref7):
    """Round 5: ltp_sample_batch flags bit 4 (sampleBatch(verdict=False)) — capped rows whose caller does not need cc:59-61's verdict
    from this call: the walk kernels stop at the cap. Rows, lengths and every other status bit are those of the default call, the
    END_LIMIT bit stays clear; the verdict is still available from planSwitchTimesBatch(end_limit=True). Both kernel forms, float32."""
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.004, device=0, **lim)       # 4 ms: ~0.5 % of random plans overshoot a joint limit
    n = 20000
    qg, q0, v0, a0 = (x.clone() for x in ltp.generateQueries(n, seed=77))
    v0[5, 0] = 99.0
    for cap, f32, kw in ((64, False, {}), (16, False, {}), (16, True, dict(auto_waves=False)), (200, False, {})):
        ltp.setMaxSamples(cap)
        dt = torch.float32 if f32 else torch.float64
        b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        full = torch.full((int(b.offsets[-1].item()) + 8,), 3.0, dtype=dt, device="cuda")
        ltp.sampleBatch(b, 0, n, full, walk=True, **kw)
        st = b.status.clone()
        assert int(((st & amd.STATUS_END_LIMIT) != 0).sum().item()) >= 3
        b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        rows = torch.full_like(full, 3.0)
        ltp.sampleBatch(b2, 0, n, rows, walk=True, verdict=False, **kw)
        torch.cuda.synchronize()
        assert torch.equal(rows, full), (cap, f32)
        assert torch.equal(b2.traj_len, b.traj_len)
        assert torch.equal(b2.status, st & ~amd.STATUS_END_LIMIT), (cap, f32)
        b3 = ltp.planSwitchTimesBatch(qg, q0, v0, a0, end_limit=True)
        torch.cuda.synchronize()
        assert torch.equal(b3.status, st)


@pytest.mark.gpu
def test_sample_opts_struct_selects_the_same_kernels_as_the_flag_word(amd, ref7):
    """ltp_sample_batch_ex takes its policy as the named fields of ltp_sample_opts; each field value launches the
    kernel the packed flag word of ltp_sample_batch launched (ltp_debug last kernel name), and every choice writes the same bytes."""
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.004, device=0, **lim)
    n = 6000
    qg, q0, v0, a0 = ltp.generateQueries(n, seed=91)
    for cap, dt in ((0, torch.float64), (24, torch.float64), (128, torch.float32)):
        ltp.setMaxSamples(cap)
        b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        ref = torch.full((int(b.offsets[-1].item()) + 8,), 3.0, dtype=dt, device="cuda")
        ltp.sampleBatch(b, 0, n, ref)                      # flags = 1: non-temporal stores, everything else automatic
        auto_kernel = ltp.lastSamplerKernel()
        st = b.status.clone()
        pairs = [("auto", {}), ("fused", dict(tables=False, walk=False)), ("walk", dict(walk=True)),
                 ("walk_streaming", dict(walk=True, auto_waves=False)), ("table", dict(tables=True, walk=False))]
        for sampler, kw in pairs:
            a = torch.full_like(ref, 3.0)
            b1 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            ltp.sampleBatch(b1, 0, n, a, **kw)
            k_flags = ltp.lastSamplerKernel()
            e = torch.full_like(ref, 3.0)
            b2 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
            ltp.sampleBatchEx(b2, 0, n, e, sampler=sampler)
            torch.cuda.synchronize()
            assert ltp.lastSamplerKernel() == k_flags, (cap, sampler, ltp.lastSamplerKernel(), k_flags)
            if sampler == "auto":
                assert k_flags == auto_kernel
            assert torch.equal(e, ref) and torch.equal(a, ref), (cap, sampler)
            assert torch.equal(b2.status, st) and torch.equal(b2.traj_len, b.traj_len)
        # plain stores, an interleave factor and "no verdict" are fields too
        e = torch.full_like(ref, 3.0)
        b3 = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
        ltp.sampleBatchEx(b3, 0, n, e, nontemporal=False, interleave=7, verdict=False)
        torch.cuda.synchronize()
        assert torch.equal(e, ref) and "_nt" not in ltp.lastSamplerKernel().replace("k_sample_walk", "")


@pytest.mark.gpu
def test_envelope_default_is_the_analytic_form_and_has_the_exhaustive_bits(amd):
    """the default envelope mode is LTP_ENVELOPE_ANALYTIC since round 6 (profiles/r06_envelope_mode_soak.json: 8.8e9
    window values, none different from the exhaustive form); the exhaustive form stays available and, here too, has the same bits."""
    import torch
    D, lim = amd.limit_set("panda")
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    q = ltp.generateQueries(30000, seed=5)
    b = ltp.planSwitchTimesBatch(*q)
    dflt = ltp.envelopeBatch(b, 0, 30000, 48, 40).clone()
    assert ltp.lastSamplerKernel() == "k_envelope_walk analytic"
    ltp.setEnvelopeMode("exhaustive")
    b2 = ltp.planSwitchTimesBatch(*q)
    ex = ltp.envelopeBatch(b2, 0, 30000, 48, 40)
    torch.cuda.synchronize()
    assert "analytic" not in ltp.lastSamplerKernel()
    assert torch.equal(torch.nan_to_num(dflt, nan=7.0), torch.nan_to_num(ex, nan=7.0)) and torch.equal(b.status, b2.status)


@pytest.mark.gpu
def test_limit_power_tables_follow_set_limits_and_the_pow_rule(amd):
    """Round 6: the powers of a joint's limits (LimPow: a_max / j_max, its cube and fourth power, a_max^3, a_max^4, j_max^3, j_max^4) are
    formed once per ltp_create / ltp_set_limits under BOTH pow rules and picked by the handle's rule at launch. A handle that has had
    other limits and the other rule before must give the bits of a fresh one: batched switching times (the tj memo and the stored
    powers in k_opt_fast / k_reduce_scale / the queue kernels), the single call, and the one-joint entry points."""
    import torch
    D, lim_a = amd.limit_set("panda")
    _, lim_b = amd.limit_set("ref")
    n = 40000
    used = amd.LongTermPlanner(D, 0.001, device=0, **lim_b)
    q_b = used.generateQueries(n, seed=3)
    used.setPowRule("exact")
    used.planSwitchTimesBatch(*q_b)                                # the handle has run with other limits and the other rule
    used.setLimits(**lim_a)
    for rule in ("libm", "exact", "libm"):
        used.setPowRule(rule)
        fresh = amd.LongTermPlanner(D, 0.001, device=0, **lim_a)
        fresh.setPowRule(rule)
        q = fresh.generateQueries(n, seed=12)
        bu, bf = used.planSwitchTimesBatch(*q), fresh.planSwitchTimesBatch(*q)
        torch.cuda.synchronize()
        for name in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status"):
            x, y = getattr(bu, name), getattr(bf, name)
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)), (rule, name)
        qh = [t[:3].cpu().numpy() for t in q]
        su, sf = used.planBatchHost(*qh, sample=False), fresh.planBatchHost(*qh, sample=False)       # k_plan_small
        assert su["t_scaled"].tobytes() == sf["t_scaled"].tobytes() and su["t_opt"].tobytes() == sf["t_opt"].tobytes()
        # and they are the bits of the batched path (same device functions, with and without the memo)
        assert su["t_scaled"].tobytes() == bf.t_scaled[:3].cpu().numpy().tobytes()
        ou = used.optSwitchTimes(2, float(qh[0][0, 2]), float(qh[1][0, 2]), float(qh[2][0, 2]), float(qh[3][0, 2]), lim_a["v_max"][2])
        of = fresh.optSwitchTimes(2, float(qh[0][0, 2]), float(qh[1][0, 2]), float(qh[2][0, 2]), float(qh[3][0, 2]), lim_a["v_max"][2])
        assert np.asarray(ou[1]).tobytes() == np.asarray(of[1]).tobytes()


@pytest.mark.gpu
def test_sample_opts_from_an_older_header_keep_the_newer_fields_at_their_defaults(amd, ref7):
    """ltp_sample_opts is size-versioned: a caller built against a SHORTER struct (here: size, format, stores, sampler = 16 bytes) is
    served as if the fields it does not know were zero, whatever bytes follow its struct in memory."""
    import ctypes as C
    import torch
    D, lim, _, _ = ref7
    ltp = amd.LongTermPlanner(D, 0.004, device=0, **lim)
    ltp.setMaxSamples(48)
    n = 3000
    q = ltp.generateQueries(n, seed=17)
    b = ltp.planSwitchTimesBatch(*q)
    ref = torch.full((int(b.offsets[-1].item()) + 8,), 3.0, dtype=torch.float64, device="cuda")
    ltp.sampleBatchEx(b, 0, n, ref, sampler="walk")
    st = b.status.clone()
    raw = (C.c_int * 7)(16, 0, 0, 2, 0x7fffffff, -5, 99)          # size 16: [verdict, interleave, dry_run] are not the caller's
    b2 = ltp.planSwitchTimesBatch(*q)
    got = torch.full_like(ref, 3.0)
    rec = b2.c_records()
    rc = ltp._lib.ltp_sample_batch_ex(ltp._h, 0, n, C.byref(b2.queries), C.byref(rec), b2.offsets.data_ptr(), got.data_ptr(), got.numel(),
                                      C.addressof(raw), None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(got, ref) and torch.equal(b2.status, st)
    raw[0] = 28                                                    # the full struct: now the garbage IS the caller's, and is refused
    assert ltp._lib.ltp_sample_batch_ex(ltp._h, 0, n, C.byref(b2.queries), C.byref(rec), b2.offsets.data_ptr(), got.data_ptr(), got.numel(),
                                        C.addressof(raw), None) == 1
