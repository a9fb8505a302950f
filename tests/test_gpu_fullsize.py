"""BASELINE.json's full size (1M x 7-DoF, 1 ms, full sampling) checked through size-independent properties.

The oracle cannot sample 386 GB in test time, so at this size the sampler is checked by what must hold for ANY
correct output of getTrajectory (cc:706-841): the Euler recurrences between the stored arrays, exact zero end
velocity/acceleration, goal reached within the reference's own tolerance, a chunking-independent checksum of every
byte, and record invariants of stages 1-3; a strided subsample is compared with the oracle directly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [("panda", 1_000_000, 40, 24), ("ref30", 1_000_000, 160, 96)]   # (limit set, queries, tile GiB of pass 1 / pass 2): BASELINE.json configs[2] and configs[4] at full size


def _chunks(off, cap):
    first, n = 0, off.size - 1
    while first < n:
        end = int(np.searchsorted(off, off[first] + np.uint64(cap), side="right")) - 1
        assert end > first
        yield first, min(end, n)
        first = min(end, n)


@pytest.mark.parametrize("name,N,gib1,gib2", CASES)
def test_full_size_batch(oracle_mod, name, N, gib1, gib2):
    # panda: BASELINE.json config 3 (1M x 7-DoF); ref30: the 30-DoF humanoid case of config 5 (joint loop over 4 rounds
    # of 8 joint-waves, 4 joint groups per plan in the sampler, ~31 % of joints on the modified jerk profile)
    import torch
    import longtermplanner_amd as amd
    D, lim = amd.limit_set(name)
    Ts = 0.001
    ltp = amd.LongTermPlanner(D, Ts, device=0, **lim)
    qg, q0, v0, a0 = ltp.generateQueries(N, seed=12345)
    b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
    torch.cuda.synchronize()

    # ---- record invariants (stages 1-3) ----
    ran = (b.status & 7) == 0
    assert ran.float().mean().item() > 0.999
    ts = b.t_scaled[ran]
    # ordered up to the reference's own slack: optBraking keeps a phase-2 duration in (-t_sample, 0) unclamped (cc:685)
    assert torch.all(ts >= -Ts) and torch.all(ts[..., 1:] >= ts[..., :-1] - Ts), "switch times must be ordered (up to t_sample)"
    idx = torch.arange(N, device="cuda")[ran]
    slow = b.slowest[ran].long()
    assert torch.equal(b.t_required[ran], b.t_opt[idx, slow, 6]), "t_required is the slowest joint's optimal end time"
    assert torch.equal(b.t_scaled[idx, slow], b.t_opt[idx, slow]), "the slowest joint keeps its optimal times"
    assert torch.all(b.t_opt[ran][..., 6] <= b.t_required[ran][:, None]), "no joint is slower than the slowest"
    lag = b.t_required[ran][:, None] - ts[..., 6]
    assert torch.all(lag > -0.01 - 1e-12), "timeScaling's window: never more than tol/10 late"
    assert (lag < 0.1).float().mean().item() > 0.995, "all but the rare fallback joints are synchronised within tol"
    want_len = torch.ceil(ts[..., 6] / Ts).max(dim=1).values.int() + 1
    assert torch.equal(b.traj_len[ran], want_len)
    assert torch.all(b.traj_len[~ran] == 0)
    stride = (b.traj_len.long() + 31) // 32 * 32
    off = b.offsets.cpu().numpy().view(np.uint64)
    assert np.array_equal(np.diff(off.astype(np.int64)), (4 * D * stride).cpu().numpy())

    # ---- a strided subsample against the oracle, bit for bit on the integers, 1e-9 on the times ----
    sub = np.arange(0, N, 997 if D == 7 else 4999)
    host = [x.cpu().numpy()[sub] for x in (qg, q0, v0, a0)]
    o = oracle_mod.Oracle(D, Ts, **lim).plan_batch(*host, sample=False)
    assert np.array_equal(b.traj_len.cpu().numpy()[sub], o["traj_len"])
    assert np.array_equal(b.slowest.cpu().numpy()[sub], o["slowest"])
    assert np.array_equal(b.mod.cpu().numpy()[sub], o["mod"])
    assert np.nanmax(np.abs(b.t_scaled.cpu().numpy()[sub] - o["t_scaled"])) <= 1e-9

    # ---- the sampler, twice with different chunking / block order / store flavour ----
    lens = b.traj_len.long()
    checks = []
    for cap_gib, spread, streaming in ((gib1, 0, True), (gib2, 1, False)):
        tile = torch.empty(cap_gib * (1 << 30) // 8, dtype=torch.float64, device="cuda")
        checksum = torch.zeros((), dtype=torch.int64, device="cuda")
        goal_err = torch.zeros(N, dtype=torch.float64, device="cuda")      # per plan: max over joints of |q_end - q_goal|
        q_end = torch.zeros((N, D), dtype=torch.float64, device="cuda")
        n_plans = 0
        for first, end in _chunks(off, tile.numel()):
            used = int(off[end] - off[first])
            tile[:used].zero_()                      # padding is never written: zero it so the checksum sees only samples
            ltp.sampleBatch(b, first, end - first, tile, streaming=streaming, spread=spread)
            checksum += tile[:used].view(torch.int64).sum()    # exact, order-independent (mod 2^64)
            sel = torch.arange(first, end, device="cuda")[lens[first:end] > 0]
            base = torch.from_numpy((off[first:end] - off[first]).astype(np.int64)).cuda()[lens[first:end] > 0]
            L, S = lens[sel], stride[sel]
            rows = base[:, None] + torch.arange(D, device="cuda")[None, :] * S[:, None]          # q rows
            last = rows + (L - 1)[:, None]
            q_last, v_last, a_last = tile[last], tile[last + (D * S)[:, None]], tile[last + (2 * D * S)[:, None]]
            assert torch.all(v_last == 0.0) and torch.all(a_last == 0.0), "final velocity/acceleration are set to exactly 0"
            goal_err[sel] = (q_last - qg[sel]).abs().max(dim=1).values
            q_end[sel] = q_last
            n_plans += sel.numel()
            # sample 0 is the state after one Euler step from (q_0, v_0, a_0) (cc:810-812)
            j_first, a_first = tile[rows + (3 * D * S)[:, None]], tile[rows + (2 * D * S)[:, None]]
            v_first, q_first = tile[rows + (D * S)[:, None]], tile[rows]
            assert torch.allclose(a_first, a0[sel] + Ts * j_first, rtol=0, atol=1e-12)
            assert torch.allclose(v_first, v0[sel] + Ts * a_first, rtol=0, atol=1e-12)
            assert torch.allclose(q_first, q0[sel] + Ts * v_first, rtol=0, atol=1e-12)
            # q[i] = q[i-1] + Ts*v[i] holds at EVERY sample (snaps only ever touch v and a), cc:830
            for p in sel[:: max(1, sel.numel() // 40)].tolist():
                o0 = int(off[p] - off[first])
                Lp, Sp = int(lens[p]), int(stride[p])
                blk = tile[o0:o0 + 4 * D * Sp].view(4, D, Sp)[:, :, :Lp]
                assert (blk[0, :, 1:] - (blk[0, :, :-1] + Ts * blk[1, :, 1:])).abs().max().item() < 1e-12
                # a[i] = a[i-1] + Ts*j[i] up to the last switch, then exactly 0 (cc:815-820)
                da = (blk[2, :, 1:] - (blk[2, :, :-1] + Ts * blk[3, :, 1:])).abs()
                assert torch.all((da < 1e-11) | (blk[2, :, 1:] == 0.0))
        checks.append(int(checksum.item()))
        assert n_plans == int(ran.sum().item())
        # Goal reached: the reference's integration tests allow 0.02 on its own (soft) limit set; with the stiff panda
        # limits SURVEY.md App. B measured mean 9.1e-4 / max 4.2e-2 over 2 000 plans of the reference itself. The
        # bulk must be that good here; the worst plans of the million are the ALGORITHM's outliers, which is shown
        # by reproducing their end positions with the oracle to 1e-9.
        ge = goal_err[ran]
        assert ge.mean().item() < 4e-3 and torch.quantile(ge[:: 16], 0.999).item() < 0.03   # per plan: max over the joints
        worst = torch.topk(goal_err, 8 if D == 7 else 2).indices.cpu().numpy()
        orc = oracle_mod.Oracle(D, Ts, **lim)
        for p in worst:
            r = orc.plan_trajectory(*[x[p].cpu().numpy() for x in (qg, q0, v0, a0)])
            assert r["status"] in (1, 2) and r["length"] == int(lens[p])
            assert np.max(np.abs(r["q"][:, -1] - q_end[p].cpu().numpy())) <= 1e-9
        del tile
        torch.cuda.empty_cache()
    assert checks[0] == checks[1], "every byte of every trajectory must be independent of chunking, block order and store flavour"
    end_limit = ((b.status & amd.STATUS_END_LIMIT) != 0).float().mean().item()
    assert end_limit < 0.01


@pytest.mark.parametrize("name,N,cap,f32", [("panda", 1_000_000, 64, False), ("panda", 1_000_000, 256, True), ("ref30", 300_000, 128, False)])
def test_full_size_short_rows_table_pass_equals_fused_sampler(name, N, cap, f32):
    """The short-row sampler (packed run tables from the table pass, expanded by the loader wave; several joints per streaming
    wave for rows of at most 32 slots) against the fused sampler at BASELINE's batch size: every stored byte and every status
    must be equal, and so must the envelopes, whose tables take the same packed route."""
    import torch
    import longtermplanner_amd as amd
    D, lim = amd.limit_set(name)
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    q = ltp.generateQueries(N, seed=777)
    q[1][123456 % N, 0] = 99.0                                   # a rejected plan
    ltp.setMaxSamples(cap)
    got = {}
    for mode in ("fused", "tables", "default"):
        b = ltp.planSwitchTimesBatch(*q)
        total = int(b.offsets[-1].item())
        tile = torch.full((total,), 7.0, dtype=torch.float32 if f32 else torch.float64, device="cuda")
        kw = {} if mode == "default" else dict(tables=(mode == "tables"), walk=False)
        ltp.sampleBatch(b, 0, N, tile, **kw)
        torch.cuda.synchronize()
        kern = ltp.lastSamplerKernel()
        # what the library takes by itself for capped rows: the walk kernel (tables kept in the compute unit) up to 63 joints, the
        # table pass beyond
        if mode == "default":
            assert kern.startswith("k_sample_walk" if D <= 63 else "k_sample_tab"), kern
        else:
            assert kern.startswith("k_sample_tab") == (mode == "tables"), (mode, kern)
        got[mode] = (tile, b.status.clone(), b.traj_len.clone())
    for mode in ("tables", "default"):
        for a, w in zip(got[mode], got["fused"]):
            assert torch.equal(a, w), mode
    assert int((got["fused"][1] == 0).sum().item()) > 0.99 * N
    del got
    ltp.setMaxSamples(0)
    b = ltp.planSwitchTimesBatch(*q)
    env = {}
    for mode, flag in (("fused", -1), ("tables", 1)):
        ltp.setTablePass(flag)
        env[mode] = ltp.envelopeBatch(b, 0, N, 64, 16).clone()
    torch.cuda.synchronize()
    assert torch.equal(env["tables"].nan_to_num(5.0), env["fused"].nan_to_num(5.0))


def test_full_size_receding_horizon_through_rows_walk_kernel_equals_fused():
    """Ten receding-horizon cycles of 1 M plans through 128-sample rows, as bench.py --receding 10:100 --max-samples 128 runs them:
    every cycle's rows (k_sample_walk by default; plans that restart mid-motion and trajectories that end inside the cap bring wide batches)
    must equal the fused sampler's rows of the same cycle bit for bit, and so must the restart states."""
    import torch
    import longtermplanner_amd as amd
    D, lim = amd.limit_set("panda")
    N = 1_000_000
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    qg, q0, v0, a0 = ltp.generateQueries(N, seed=4711)
    ltp.setMaxSamples(128)
    s0, s1, s2 = q0, v0, a0
    wide_cycles = 0
    for cycle in range(10):
        b = ltp.planSwitchTimesBatch(qg, s0, s1, s2)
        total = int(b.offsets[-1].item())
        t_walk = torch.full((total,), 7.0, dtype=torch.float64, device="cuda")
        t_fused = torch.full((total,), 7.0, dtype=torch.float64, device="cuda")
        ltp.sampleBatch(b, 0, N, t_walk)
        assert ltp.lastSamplerKernel().startswith("k_sample_walk")
        st_walk = b.status.clone()
        b2 = ltp.planSwitchTimesBatch(qg, s0, s1, s2)
        ltp.sampleBatch(b2, 0, N, t_fused, tables=False, walk=False)
        assert ltp.lastSamplerKernel() == "k_sample"
        torch.cuda.synchronize()
        assert torch.equal(t_walk, t_fused), cycle
        assert torch.equal(st_walk, b2.status) and torch.equal(b.traj_len, b2.traj_len), cycle
        wide_cycles += int(((b.traj_len > 0) & (b.traj_len <= 128)).sum().item())
        s0, s1, s2 = ltp.replanStates(b, 0, N, t_walk, 100)
        del t_walk, t_fused
    assert wide_cycles >= 1000, "hardly any trajectory ended inside the cap: the wide batches were not exercised at scale"
