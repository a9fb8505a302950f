"""The on-device consumer HOOK: ltp_build_tables_batch + include/ltp_run_tables.hpp.

A USER-side consumer (tests/cpp/example_consumer.hip, compiled by plain hipcc against the public device header only) reduces
trajectories it never sees as rows. Its results are checked against the same reductions of
  * the dense rows ltp_sample_batch writes for the same batch — bit for bit (maxima, minima and counts are order-free), and
  * the CPU oracle's dense rows (cc:706-841) — to 1e-9, counts up to samples that sit within 1e-9 of the threshold.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-9


@pytest.fixture(scope="module")
def amd():
    import longtermplanner_amd as m
    return m


@pytest.fixture(scope="module")
def consumer(amd):
    amd.LongTermPlanner                                    # (loads torch's HIP runtime first: one runtime per process)
    from longtermplanner_amd import _abi
    _abi.lib()
    path = os.path.join(ROOT, "tests", "cpp", "libexample_consumer.so")
    assert os.path.exists(path), "tests/cpp/libexample_consumer.so is missing: run __graft_entry__.build()"
    lib = C.CDLL(path)
    lib.example_peak_velocity.restype = C.c_int
    lib.example_peak_velocity.argtypes = [C.c_void_p, C.c_longlong, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.example_box_clearance.restype = C.c_int
    lib.example_box_clearance.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def _dense_rows(ltp, batch, first, count, D):
    """rows of plans [first, first+count) as a list of [4][D][len] torch views into one tile (None where nothing was sampled)"""
    import torch
    off = batch.offsets.cpu().numpy().view(np.uint64)
    total = int(off[first + count] - off[first])
    tile = torch.zeros(max(total, 2), dtype=torch.float64, device="cuda")
    ltp.sampleBatch(batch, first, count, tile)
    torch.cuda.synchronize()
    lens = batch.traj_len.cpu().numpy()
    st = batch.status.cpu().numpy()
    rows = []
    for p in range(first, first + count):
        n = int(lens[p])
        if n <= 0 or (int(st[p]) & 0x77):
            rows.append(None)
            continue
        stride = (n + 31) // 32 * 32
        base = int(off[p] - off[first])
        rows.append(tile[base:base + 4 * D * stride].view(4, D, stride)[:, :, :n])
    return rows, tile


@pytest.mark.parametrize("limits,ts,n,first", [("panda", 0.001, 3000, 0), ("ref", 0.004, 1500, 777), ("ref30", 0.002, 400, 13)])
def test_user_consumer_peak_velocity_equals_the_reduced_rows(amd, oracle_mod, consumer, limits, ts, n, first):
    """Lane form (RunTableView + for_each_sample), any dof, a range that does not start at plan 0."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    dq = ltp.generateQueries(n, seed=31)
    batch = ltp.planSwitchTimesBatch(*dq)
    count = n - first
    tables = ltp.buildRunTables(batch, first, count)
    lanes = count * D
    thr = 0.35 * float(min(lim["v_max"]))
    mx = torch.full((lanes,), -7.0, dtype=torch.float64, device="cuda")
    at = torch.full((lanes,), -7, dtype=torch.int32, device="cuda")
    above = torch.full((lanes,), -7, dtype=torch.int64, device="cuda")
    assert consumer.example_peak_velocity(tables.data_ptr(), lanes, ts, thr, mx.data_ptr(), at.data_ptr(), above.data_ptr(), None) == 0
    torch.cuda.synchronize()
    st_tables = batch.status.cpu().numpy().copy()              # END_LIMIT set by the table pass ...
    rows, tile = _dense_rows(ltp, batch, first, count, D)
    assert np.array_equal(batch.status.cpu().numpy(), st_tables), "... is the sampler's verdict"
    mx, at, above = mx.cpu().numpy().reshape(count, D), at.cpu().numpy().reshape(count, D), above.cpu().numpy().reshape(count, D)
    host = [x.cpu().numpy() for x in dq]
    orc = oracle_mod.Oracle(D, ts, **lim)
    o = orc.plan_batch(*host, sample=False)
    checked_oracle = 0
    for i, r in enumerate(rows):
        if r is None:
            assert np.all(mx[i] == -1.0) and np.all(at[i] == -1) and np.all(above[i] == 0)
            continue
        av = r[1].abs()
        want_mx, want_at = av.max(dim=1)
        assert mx[i].tobytes() == want_mx.cpu().numpy().tobytes(), (i, mx[i], want_mx)
        first_at = (av == want_mx[:, None]).to(torch.int32).argmax(dim=1).cpu().numpy()       # first sample holding the maximum
        assert np.array_equal(at[i], first_at), i
        assert np.array_equal(above[i], (av > thr).sum(dim=1).cpu().numpy()), i
        if i % 37 == 0 and checked_oracle < 25:                # the oracle's own rows (cc:706-841)
            p = first + i
            ln, oq, ov, oa, oj = orc.get_trajectory(o["t_scaled"][p], o["dir"][p], o["mod"][p], host[1][p], host[2][p], host[3][p], o["v_drive"][p])
            assert ln == r.shape[2]
            oav = np.abs(ov)
            assert np.max(np.abs(oav.max(axis=1) - mx[i])) < TOL
            sure = np.abs(oav - thr) > TOL                      # samples that cannot flip within the tolerance
            lo_cnt = (sure & (oav > thr)).sum(axis=1)
            assert np.all(above[i] >= lo_cnt) and np.all(above[i] <= lo_cnt + (~sure).sum(axis=1))
            checked_oracle += 1
    assert checked_oracle >= 5
    del tile


@pytest.mark.parametrize("limits,ts,n", [("panda", 0.001, 2500), ("ref", 0.002, 1200)])
def test_user_consumer_box_clearance_equals_the_reduced_rows(amd, oracle_mod, consumer, limits, ts, n):
    """Block form (fetch_run_tables + install_run_tables + RunCursor in LDS): a reduction over the joint VECTOR per sample."""
    import torch
    D, lim = amd.limit_set(limits)
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    dq = ltp.generateQueries(n, seed=77)
    batch = ltp.planSwitchTimesBatch(*dq)
    tables = ltp.buildRunTables(batch, 0, n)
    mid = 0.5 * (np.asarray(lim["q_min"]) + np.asarray(lim["q_max"]))
    half = 0.12 * (np.asarray(lim["q_max"]) - np.asarray(lim["q_min"]))
    lo = torch.tensor(mid - half, dtype=torch.float64, device="cuda")
    hi = torch.tensor(mid + half, dtype=torch.float64, device="cuda")
    clear = torch.full((n,), -7.0, dtype=torch.float64, device="cuda")
    at = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    assert consumer.example_box_clearance(tables.data_ptr(), n, D, ts, lo.data_ptr(), hi.data_ptr(), clear.data_ptr(), at.data_ptr(), None) == 0
    torch.cuda.synchronize()
    rows, tile = _dense_rows(ltp, batch, 0, n, D)
    clear, at = clear.cpu().numpy(), at.cpu().numpy()
    host = [x.cpu().numpy() for x in dq]
    orc = oracle_mod.Oracle(D, ts, **lim)
    o = orc.plan_batch(*host, sample=False)
    entered, checked_oracle = 0, 0
    for i, r in enumerate(rows):
        if r is None:
            assert np.isnan(clear[i]) and at[i] == -1
            continue
        q = r[0]                                               # [D][len]
        d = torch.clamp(torch.maximum(lo[:, None] - q, q - hi[:, None]).max(dim=0).values, min=0.0)
        want = d.min()
        assert clear[i].tobytes() == want.cpu().numpy().tobytes(), (i, clear[i], float(want))
        assert at[i] == int((d == want).to(torch.int32).argmax()), i
        entered += int(clear[i] == 0.0)
        if i % 41 == 0 and checked_oracle < 20:
            ln, oq, ov, oa, oj = orc.get_trajectory(o["t_scaled"][i], o["dir"][i], o["mod"][i], host[1][i], host[2][i], host[3][i], o["v_drive"][i])
            od = np.maximum(np.maximum(lo.cpu().numpy()[:, None] - oq, oq - hi.cpu().numpy()[:, None]).max(axis=0), 0.0)
            assert abs(od.min() - clear[i]) < TOL
            checked_oracle += 1
    assert checked_oracle >= 5 and 0 < entered < n               # the box is entered by some trajectories and missed by others
    del tile


def test_build_tables_argument_checks(amd):
    import torch
    D, lim = amd.limit_set("panda")
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    dq = ltp.generateQueries(100, seed=3)
    batch = ltp.planSwitchTimesBatch(*dq)
    assert int(ltp._lib.ltp_run_tables_bytes(ltp._h, 100)) == ((100 * D + 63) // 64) * 114 * 64 * 8
    small = torch.empty(16, dtype=torch.int64, device="cuda")
    with pytest.raises(amd.LtpError):
        ltp.buildRunTables(batch, 0, 100, out=small[:0].new_empty(8))            # too small
    ltp.setSampleTime(0.002)                                                     # geometry changed since the batch was planned
    with pytest.raises(amd.LtpError):
        ltp.buildRunTables(batch, 0, 100)
