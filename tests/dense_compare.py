"""Dense-trajectory comparison of the HIP path with the CPU oracle — TEST INFRASTRUCTURE (used by tests/test_gpu_parity.py and
tools/dense_soak.py; imports the oracle, so nothing in the product may import this).

soak(): a batch is planned and sampled on the device chunk by chunk (a reused tile, as bench.py does), each chunk's rows are copied
to pinned host memory, and host threads run the oracle's full planTrajectory (cc:7-63) on the same queries and compare EVERY
sample (oracle/ltp_oracle.c: ltpo_compare_dense). Plans beyond the tolerance are listed with a cause class, as SURVEY.md §8(d)
asks: root-classification, window-test flip, sample-index flip, else rounding — and tested against the budget a jerk-correction
sample is allowed (below).
"""
import ctypes as C
import json
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

TOL = 1e-9
THREADS = max(1, min(len(os.sched_getaffinity(0)), 32))
CHUNK_BYTES = 3 << 30                       # per host buffer; two buffers in flight

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_ulonglong)


def fuzz_limits(rng, trial, wide):
    """One fuzzed limit set: dof 1-12, Ts 1-10 ms, random per-joint limits; wide: also Ts down to 0.1 ms, j_max / Ts up to 1e9
    (every third set) and slow-jerk sets whose trajectories have 1e4-1e5 samples (every third set)."""
    D = int(rng.integers(1, 13))
    ts = float(rng.choice([0.0001, 0.00025, 0.001, 0.002, 0.004, 0.01] if wide else [0.001, 0.002, 0.004, 0.01]))
    v_max = rng.uniform(0.5, 3.0, D)
    a_max = rng.uniform(1.0, 20.0, D)
    j_max = a_max * rng.uniform(5.0, 600.0, D)
    if wide and trial % 3 == 1:
        j_max = np.minimum(a_max * rng.uniform(500.0, 5000.0, D), 1e9 * ts)       # j_max / Ts up to 1e9
    elif wide and trial % 3 == 2:
        j_max = a_max * rng.uniform(0.05, 2.0, D)                                  # slow jerk: 1e4-1e5 samples per trajectory
    q_hi = rng.uniform(1.0, 3.5, D)
    return D, ts, dict(q_min=list(-q_hi), q_max=list(q_hi), v_max=list(v_max), a_max=list(a_max), j_max=list(j_max))


def pinned_buffers():
    import torch
    return [torch.empty(CHUNK_BYTES // 8, dtype=torch.float64, pin_memory=True) for _ in range(2)]


def cause(ltp, orc, q):
    """Why this query differs: compare the device's records with the oracle's (SURVEY.md §8(d) classes)."""
    dev = ltp.planBatchHost(*q, sample=False)
    o = orc.plan_batch(*q, sample=False)
    if np.any(dev["mod"][0] != o["mod"][0]) or np.any(dev["dir"][0] != o["dir"][0]) or dev["slowest"][0] != o["slowest"][0]:
        dv = np.abs(dev["v_drive"][0] - o["v_drive"][0])
        return "root-classification" if np.any(np.isfinite(dv) & (dv > 1e-6)) else "window-test flip"
    if dev["traj_len"][0] != o["traj_len"][0]:
        return "sample-index flip (trajectory length)"
    Ts = ltp.t_sample
    a, b = dev["t_scaled"][0] / Ts, o["t_scaled"][0] / Ts
    if np.any(np.floor(a) != np.floor(b)) or np.any(np.ceil(a) != np.ceil(b)):
        return "sample-index flip"
    dt = float(np.nanmax(np.abs(dev["t_scaled"][0] - o["t_scaled"][0])))
    gain = float(np.max(orc.j_max)) / Ts
    if dt > 0.0:
        # cc:771-807: a fractional jerk sample is (t - Ts floor(t / Ts)) / Ts * j_max — a last-bit difference of a switching time
        # (pow's last bit, which also differs between libm builds) reaches that ONE jerk sample multiplied by j_max / Ts
        return (f"rounding: switching times differ by {dt:.2e} (last bits); the fractional jerk samples (cc:771-807) carry that "
                f"times j_max / Ts = {gain:.1e}")
    return "rounding"


def soak(name, D, lim, Ts, n, seed, bufs, exact=False, quiet=False, matlab=False, pow_rule="exact"):
    """exact: compare against the oracle's DIAGNOSTIC exact-pow twin instead of the libm oracle (the parity reference).
    matlab: both sides in MATLAB semantics (LTPlanner.m; the device's rows then come from k_sample_walk_matlab_*).
    pow_rule: the device's pow rule (ltp_set_pow_rule): "libm" forms the powers as the libm oracle's pow does."""
    import torch
    import longtermplanner_amd as amd
    import oracle
    olib = oracle.lib(exact)
    ltp = amd.LongTermPlanner(D, Ts, device=0, **lim)
    orc = oracle.Oracle(D, Ts, exact_pow=exact, semantics="matlab" if matlab else "cpp", **lim)
    if matlab:
        ltp.setSemantics("matlab")
    ltp.setPowRule(pow_rule)
    t0 = time.time()
    dq = ltp.generateQueries(n, seed=seed)
    b = ltp.planSwitchTimesBatch(*dq)
    torch.cuda.synchronize()
    host = [np.ascontiguousarray(x.cpu().numpy()) for x in dq]
    off = b.offsets.cpu().numpy().view(np.uint64)
    cap = CHUNK_BYTES // 8
    tile = torch.empty(cap, dtype=torch.float64, device="cuda")
    maxd = np.zeros((n, 4))
    flag = np.zeros(n, dtype=np.int32)
    compared = [0]
    pending = [None, None]
    bounds, first = [], 0
    while first < n:
        end = int(np.searchsorted(off, off[first] + np.uint64(cap), side="right")) - 1
        assert end > first, "one trajectory does not fit the chunk buffer"
        bounds.append((first, min(end, n)))
        first = min(end, n)
    with ThreadPoolExecutor(THREADS) as ex:
        # the device's lengths / status words are complete only after a chunk's sampler ran (END_LIMIT bits): per chunk copies
        for ci, (f, e) in enumerate(bounds):
            slot = ci & 1
            if pending[slot] is not None:
                for fut in pending[slot]:
                    compared[0] += fut.result()
            used = int(off[e] - off[f])
            tile[:used].zero_()
            ltp.sampleBatch(b, f, e - f, tile)
            hb = bufs[slot]
            hb[:used].copy_(tile[:used], non_blocking=True)
            torch.cuda.synchronize()
            dev_len = np.ascontiguousarray(b.traj_len[f:e].cpu().numpy())
            dev_st = np.ascontiguousarray(b.status[f:e].cpu().numpy())
            hnp = hb.numpy()
            step = -(-(e - f) // THREADS)

            # (everything a task reads is bound here: tasks of this chunk may still be queued when the loop variables move on — an
            # unbound `step` let late-starting tasks of small multi-chunk sets cover the wrong plan ranges, found in round 4 through
            # values_compared differing between two runs of one set; the visited check below now proves full coverage)
            def work(t, f=f, e=e, hnp=hnp, dev_len=dev_len, dev_st=dev_st, step=step):
                lo = f + t * step
                cnt = min(step, e - lo)
                if cnt <= 0:
                    return 0
                # the comparator indexes inputs / offsets / device records by absolute plan number: shift the chunk-local arrays
                return int(olib.ltpo_compare_dense(orc._ref, C.c_long(lo), C.c_long(cnt), *[x.ctypes.data_as(_dp) for x in host],
                                                   hnp.ctypes.data_as(_dp), off.ctypes.data_as(_up), C.c_ulonglong(int(off[f])),
                                                   C.cast(dev_len.ctypes.data - 4 * f, _ip), C.cast(dev_st.ctypes.data - 4 * f, _ip),
                                                   maxd[lo:].ctypes.data_as(_dp), flag[lo:].ctypes.data_as(_ip)))
            pending[slot] = [ex.submit(work, t) for t in range(THREADS)]
        for slot in (0, 1):
            if pending[slot] is not None:
                for fut in pending[slot]:
                    compared[0] += fut.result()
    assert np.all((flag & 16) != 0), f"{int(np.sum((flag & 16) == 0))} plans were never visited by the comparator"
    lens_host = b.traj_len.cpu().numpy().astype(np.int64)
    expect = int(4 * D * np.sum(lens_host[((flag & 3) == 0) & ((b.status.cpu().numpy() & 0x57) == 0)]))
    assert compared[0] == expect, (compared[0], expect, "every sample of every accepted plan exactly once")
    worst = maxd.max(axis=0)
    beyond = np.nonzero((maxd.max(axis=1) > TOL) | ((flag & 7) != 0))[0]
    outliers = []
    explained = 0
    gain = float(np.max(orc.j_max)) / Ts
    for p in beyond[:200]:
        q = [x[p:p + 1] for x in host]
        o = {"query": int(p), "max_abs_d": [float(x) for x in maxd[p]], "flags": int(flag[p]) & 15, "cause": cause(ltp, orc, q)}
        # the budget of tests/test_gpu_parity.py: an a / j sample beyond 1e-9 is explained when q and v hold, the verdicts agree and the
        # plan's switching times differ by |dt| <= 1e-9 with |d j| <= 4 |dt| j_max / Ts (a: the same integrated once: 4 |dt| j_max).
        # The factor: a jerk sample that collects corrections is a sum of terms +-(f_k / Ts) j_max, f_k = t_k - Ts floor(t_k / Ts)
        # (cc:747); the worst index (cc:798) collects f_4 J_4 + f_0 J_0 + (f_2 - f_0) J_2 = (-f_4 +- 2 f_0 -+ f_2) dir j_max / Ts —
        # four |dt|; cc:781 collects (2 f_0 - f_2): three. (Seen: exactly 2 |dt| j_max / Ts, twice in 5.3 M fuzzed plans.)
        devr, orr = ltp.planBatchHost(*q, sample=False), orc.plan_batch(*q, sample=False)
        dt = float(np.nanmax(np.abs(devr["t_scaled"][0] - orr["t_scaled"][0])))
        o["max_abs_dt"] = dt
        o["explained_by_dt_times_jmax_over_ts"] = bool((flag[p] & 7) == 0 and maxd[p][0] <= TOL and maxd[p][1] <= TOL and dt <= 1e-9 and
                                                        maxd[p][3] <= 4.0 * dt * gain and maxd[p][2] <= max(TOL, 4.0 * dt * gain * Ts))
        explained += o["explained_by_dt_times_jmax_over_ts"]
        if len(outliers) < 40:
            outliers.append(o)
    st = b.status.cpu().numpy()
    res = {"dof": D, "t_sample": Ts, "seed": seed, "device_pow_rule": pow_rule, "dense_plans": int(n), "sampled": int(np.sum((st & 0x57) == 0)),
           "values_compared": int(compared[0]), "bytes_compared": int(compared[0]) * 8,
           "max_abs_d": {k: float(worst[i]) for i, k in enumerate("qvaj")},
           "plans_beyond_tolerance": int(np.sum(maxd.max(axis=1) > TOL)),
           "verdict_mismatches": int(np.sum((flag & 1) != 0)), "length_mismatches": int(np.sum((flag & 2) != 0)),
           "end_limit_flag_mismatches": int(np.sum((flag & 4) != 0)), "end_limit_false": int(np.sum((st & 8) != 0)),
           "plans_with_bit_identical_jerk_rows": int(np.sum(((flag & 8) == 0) & ((flag & 3) == 0) & ((st & 0x57) == 0))),
           "fraction_within_tolerance": float(1.0 - beyond.size / n), "outliers_examined": int(min(beyond.size, 200)),
           "outliers_explained_by_dt": int(explained), "outliers": outliers,
           "seconds": round(time.time() - t0, 1)}
    del tile
    torch.cuda.empty_cache()
    if not quiet:
        print(name, json.dumps(res), flush=True)
    return res


