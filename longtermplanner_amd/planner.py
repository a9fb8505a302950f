"""Host-side mirror of the reference's ``LongTermPlanner`` class on top of the C ABI.

Same names, argument meaning and return conventions as
/root/reference/include/long_term_planner/long_term_planner.h:61-308 (constructor, planTrajectory,
checkInputs, setLimits, setSampleTime, setDoF and the four protected methods), plus the batched
calls that are the reason this package exists. Every method runs on the GPU through
libltp_hip.so; nothing here computes planner arithmetic on the host.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi

_dp = C.POINTER(C.c_double)


def _vec(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def _ptr(a):
    return a.ctypes.data_as(_dp)


@dataclass
class Trajectory:
    """reference struct Trajectory (long_term_planner.h:37-45): q/v/a/j are [joint][sample]."""
    dof: int = 0
    t_sample: float = 0.0
    length: int = 0
    q: List = field(default_factory=list)
    v: List = field(default_factory=list)
    a: List = field(default_factory=list)
    j: List = field(default_factory=list)


def unpack_trajectory(packed, offset, dof, length):
    """Views [dof][length] of q, v, a, j inside a packed buffer (layout: include/ltp_hip.h)."""
    stride = _abi.lib().ltp_row_stride(int(length))
    blk = packed[offset: offset + 4 * dof * stride].reshape(4, dof, stride)
    return blk[0, :, :length], blk[1, :, :length], blk[2, :, :length], blk[3, :, :length]


class DeviceBatch:
    """Device-resident records of one ltp_plan_switch_times_batch call (torch tensors)."""

    def __init__(self, n, dof, device):
        import torch
        f64 = dict(dtype=torch.float64, device=device)
        self.n, self.dof = n, dof
        self.t_opt = torch.empty((n, dof, 7), **f64)
        self.t_scaled = torch.empty((n, dof, 7), **f64)
        self.dir = torch.empty((n, dof), **f64)
        self.v_drive = torch.empty((n, dof), **f64)
        self.mod = torch.empty((n, dof), dtype=torch.int8, device=device)
        self.t_required = torch.empty((n,), **f64)
        self.slowest = torch.empty((n,), dtype=torch.int32, device=device)
        self.traj_len = torch.empty((n,), dtype=torch.int32, device=device)
        self.status = torch.empty((n,), dtype=torch.int32, device=device)
        self.offsets = torch.empty((n + 1,), dtype=torch.int64, device=device)   # uint64 on the device side

    def c_records(self):
        return _abi.Records(self.t_opt.data_ptr(), self.t_scaled.data_ptr(), self.dir.data_ptr(), self.v_drive.data_ptr(),
                            self.mod.data_ptr(), self.t_required.data_ptr(), self.slowest.data_ptr(),
                            self.traj_len.data_ptr(), self.status.data_ptr())


class LongTermPlanner:
    def __init__(self, dof=0, t_sample=0.001, q_min=(), q_max=(), v_max=(), a_max=(), j_max=(), device=0):
        # default arguments reproduce the reference's dummy constructor (long_term_planner.h:103-105)
        self._lib = _abi.lib()
        self._h = C.c_void_p()
        arrs = [_vec(x) for x in (q_min, q_max, v_max, a_max, j_max)]
        for a in arrs:
            if a.size < dof:
                raise ValueError("limit vectors need at least dof entries")
        rc = self._lib.ltp_create(int(dof), float(t_sample), *[_ptr(a) for a in arrs], int(device), C.byref(self._h))
        if rc != _abi.LTP_OK:
            raise _abi.LtpError(rc, "ltp_create failed (no HIP device? the planner has no CPU fallback)")
        self.device = int(device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.ltp_destroy(h)
            self._h = None

    def _check(self, rc):
        if rc != _abi.LTP_OK:
            raise _abi.LtpError(rc, (self._lib.ltp_last_error(self._h) or b"").decode())

    # ---- configuration (long_term_planner.h:176-205) ----
    def setLimits(self, q_min, q_max, v_max, a_max, j_max):
        arrs = [_vec(x) for x in (q_min, q_max, v_max, a_max, j_max)]
        n = min(a.size for a in arrs)
        self._check(self._lib.ltp_set_limits(self._h, n, *[_ptr(a) for a in arrs]))

    def setSampleTime(self, t_sample):
        self._check(self._lib.ltp_set_sample_time(self._h, float(t_sample)))

    def setDoF(self, dof):
        self._check(self._lib.ltp_set_dof(self._h, int(dof)))

    def setMaxSamples(self, max_samples):
        """NEW: store only the first max_samples samples of every trajectory; 0 = all (reference)."""
        self._check(self._lib.ltp_set_max_samples(self._h, int(max_samples)))

    def setSampleStride(self, stride):
        """NEW: store every stride-th sample (0, stride, 2*stride, ...); 1 = every sample (reference)."""
        self._check(self._lib.ltp_set_sample_stride(self._h, int(stride)))

    def setGoalCheck(self, enabled=True):
        """NEW, off by default: reject queries whose q_goal lies outside [q_min, q_max] up front with
        LTP_STATUS_GOAL_OUTSIDE (64) instead of planning them and failing the end-limit check (cc:59-61)."""
        self._check(self._lib.ltp_set_goal_check(self._h, 1 if enabled else 0))

    def setSemantics(self, semantics):
        """NEW: "cpp" (default: src/long_term_planner.cc, the parity reference) or "matlab" (the MATLAB original
        LTPlanner.m where it diverges: positional root picks, zeros instead of failures, no position limits, 1-based sampler with
        cumsum integration; include/ltp_hip.h LTP_SEMANTICS_MATLAB)."""
        code = {"cpp": _abi.SEMANTICS_CPP, "matlab": _abi.SEMANTICS_MATLAB}.get(semantics, semantics)
        self._check(self._lib.ltp_set_semantics(self._h, int(code)))

    def setEnvelopeMode(self, mode):
        """NEW: "analytic" (default since round 6: the samples at the ends of each run stretch and either side of the
        roots of q'(m), a few evaluations per run; identical to the exhaustive form in 8.8e9 soaked values, <= 1e-12 by construction) or
        "exhaustive" (every sample of a window: the bits of the reduced rows by construction)."""
        code = {"exhaustive": 0, "analytic": 1}.get(mode, mode)
        self._check(self._lib.ltp_set_envelope_mode(self._h, int(code)))

    def setPowRule(self, rule):
        """NEW: how the reference's pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) are formed — "libm" (default: glibc's pow restated operation
        for operation: the bits of a reference built with gcc + glibc on an FMA host; include/ltp_hip.h LTP_POW_LIBM) or "exact"
        (one rounding of the exact product, sqrt: within 1 ulp of any libm, faster stage kernels)."""
        code = {"exact": _abi.POW_EXACT, "libm": _abi.POW_LIBM}.get(rule, rule)
        self._check(self._lib.ltp_set_pow_rule(self._h, int(code)))

    def lastMatlabFlags(self):
        """MATLAB semantics: flags of the latest one-lane call (optBraking / optSwitchTimes / timeScaling): 1 = complex intermediate, 2 = error."""
        return self._lib.ltp_debug_last_matlab_flags(self._h)

    def matlabRoots(self, poly):
        """MATLAB's roots() as the MATLAB-semantics kernels compute it (device): (complex roots [n][degree] in MATLAB's order, nroots, status)."""
        poly = np.ascontiguousarray(np.atleast_2d(np.asarray(poly, dtype=np.float64)))
        n, deg = poly.shape[0], poly.shape[1] - 1
        re = np.zeros((n, deg)); im = np.zeros((n, deg)); nr = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self._check(self._lib.ltp_debug_roots_matlab_host(self._h, n, deg, _ptr(poly), _ptr(re), _ptr(im), nr.ctypes.data_as(ip), st.ctypes.data_as(ip)))
        return re + 1j * im, nr, st

    def setTablePass(self, mode, workspace_bytes=None):
        """NEW: where the run tables are built (include/ltp_hip.h, ltp_set_table_pass) — 0 automatic; 1 never the block-wide fused
        build (rows: k_sample_walk_*, envelopes: the table pass); -1 always the fused build (rows: k_sample, envelopes: in the
        kernel). Results are bit-identical either way."""
        self._check(self._lib.ltp_set_table_pass(self._h, int(mode)))
        if workspace_bytes is not None:
            self._check(self._lib.ltp_set_table_workspace(self._h, int(workspace_bytes)))

    def storedSamples(self, traj_len):
        return self._lib.ltp_stored_samples(self._h, int(traj_len))

    @property
    def dof(self):
        return self._lib.ltp_get_dof(self._h)

    @property
    def t_sample(self):
        return self._lib.ltp_get_sample_time(self._h)

    # ---- the reference's public calls ----
    def checkInputs(self, q_0, v_0, a_0):
        ok = C.c_int()
        self._check(self._lib.ltp_check_inputs_host(self._h, _ptr(_vec(q_0)), _ptr(_vec(v_0)), _ptr(_vec(a_0)), C.byref(ok)))
        return bool(ok.value)

    def planTrajectory(self, q_goal, q_0, v_0, a_0, traj: Trajectory):
        """long_term_planner.h:144-150 / src/long_term_planner.cc:7-63. `traj` is overwritten only if the
        reference would have overwritten it (status has none of the pre-sampling failure bits)."""
        r = self.planBatchHost(q_goal, q_0, v_0, a_0, sample=True)
        st = int(r["status"][0])
        if st & (_abi.STATUS_INVALID_INPUT | _abi.STATUS_OPT_FAILED | _abi.STATUS_NO_SLOWEST | _abi.STATUS_NONFINITE | _abi.STATUS_GOAL_OUTSIDE | _abi.STATUS_MATLAB_ERROR):
            return False
        n = int(r["traj_len"][0])
        q, v, a, j = unpack_trajectory(r["packed"], int(r["offsets"][0]), self.dof, n)
        traj.dof, traj.t_sample, traj.length = self.dof, self.t_sample, n
        traj.q, traj.v, traj.a, traj.j = q.copy(), v.copy(), a.copy(), j.copy()
        return (st & ~_abi.STATUS_MATLAB_COMPLEX) == 0   # the informational MATLAB bit does not make a plan fail

    # ---- the reference's protected methods (exposed to tests through a subclass there) ----
    def optBraking(self, joint, v_0, a_0, t_rel=None):
        t = np.zeros(7) if t_rel is None else np.array(t_rel, dtype=np.float64)
        q = C.c_double(); d = C.c_double()
        self._check(self._lib.ltp_opt_braking_host(self._h, int(joint), float(v_0), float(a_0), C.byref(q), _ptr(t), C.byref(d)))
        return True, q.value, t, d.value

    def optSwitchTimes(self, joint, q_goal, q_0, v_0, a_0, v_drive, t=None):
        tt = np.zeros(7) if t is None else np.array(t, dtype=np.float64)
        d = C.c_double(); m = C.create_string_buffer(1); ok = C.c_int()
        self._check(self._lib.ltp_opt_switch_times_host(self._h, int(joint), float(q_goal), float(q_0), float(v_0), float(a_0),
                                                        float(v_drive), _ptr(tt), C.byref(d), m, C.byref(ok)))
        return bool(ok.value), tt, d.value, m.raw[0]

    def timeScaling(self, joint, q_goal, q_0, v_0, a_0, dir_, t_required, scaled_t=None):
        tt = np.zeros(7) if scaled_t is None else np.array(scaled_t, dtype=np.float64)
        vd = C.c_double(); m = C.create_string_buffer(1); ok = C.c_int(); case = C.c_int()
        self._check(self._lib.ltp_time_scaling_host(self._h, int(joint), float(q_goal), float(q_0), float(v_0), float(a_0),
                                                    float(dir_), float(t_required), _ptr(tt), C.byref(vd), m, C.byref(ok),
                                                    C.byref(case)))
        return bool(ok.value), tt, vd.value, m.raw[0], case.value

    def getTrajectory(self, t, dir_, mod_jerk_profile, q_0, v_0, a_0, v_drive):
        r = self.getTrajectoryBatchHost(np.asarray(t, dtype=np.float64).reshape(1, self.dof, 7), dir_, mod_jerk_profile, q_0, v_0, a_0, v_drive)
        n = int(r["traj_len"][0])
        q, v, a, j = unpack_trajectory(r["packed"], 0, self.dof, n)
        return Trajectory(self.dof, self.t_sample, n, q.copy(), v.copy(), a.copy(), j.copy())

    # ---- batched host-pointer calls (numpy in, numpy out; synchronous) ----
    def planBatchHost(self, q_goal, q_0, v_0, a_0, sample=True):
        D = self.dof
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, D) if D else np.zeros((1, 0))) for x in (q_goal, q_0, v_0, a_0)]
        n = ins[0].shape[0]
        r = dict(t_opt=np.zeros((n, D, 7)), t_scaled=np.zeros((n, D, 7)), dir=np.zeros((n, D)), v_drive=np.zeros((n, D)),
                 mod=np.zeros((n, D), dtype=np.int8), t_required=np.zeros(n), slowest=np.zeros(n, dtype=np.int32),
                 traj_len=np.zeros(n, dtype=np.int32), status=np.zeros(n, dtype=np.int32))
        rec = _abi.Records(*[r[k].ctypes.data for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status")])
        offsets = np.zeros(n + 1, dtype=np.uint64)
        packed = _dp()
        self._check(self._lib.ltp_plan_batch_host(self._h, n, *[_ptr(x) for x in ins], C.byref(rec),
                                                  offsets.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                                  C.byref(packed) if sample else None))
        r["offsets"] = offsets
        if sample:
            total = int(offsets[n])
            r["packed"] = np.ctypeslib.as_array(packed, shape=(max(total, 1),))[:total].copy()
            self._lib.ltp_free_host(packed)
        return r

    def planEnvelopeHost(self, q_goal, q_0, v_0, a_0, window, n_windows):
        """NEW: stages 1-3 + on-device envelope consumer for numpy arrays (ltp_plan_envelope_host). Returns (records dict,
        env[n][dof][n_windows][2])."""
        D = self.dof
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, D)) for x in (q_goal, q_0, v_0, a_0)]
        n = ins[0].shape[0]
        r = dict(t_opt=np.zeros((n, D, 7)), t_scaled=np.zeros((n, D, 7)), dir=np.zeros((n, D)), v_drive=np.zeros((n, D)),
                 mod=np.zeros((n, D), dtype=np.int8), t_required=np.zeros(n), slowest=np.zeros(n, dtype=np.int32),
                 traj_len=np.zeros(n, dtype=np.int32), status=np.zeros(n, dtype=np.int32))
        rec = _abi.Records(*[r[k].ctypes.data for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status")])
        env = np.zeros((n, D, int(n_windows), 2))
        self._check(self._lib.ltp_plan_envelope_host(self._h, n, *[_ptr(x) for x in ins], int(window), int(n_windows), C.byref(rec), _ptr(env)))
        return r, env

    def getTrajectoryBatchHost(self, t, dir_, mod, q_0, v_0, a_0, v_drive):
        D = self.dof
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(-1, D, 7))
        n = t.shape[0]
        f = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(n, D)) for x in (dir_, q_0, v_0, a_0, v_drive)]
        modb = np.ascontiguousarray(np.asarray(mod, dtype=np.int8).reshape(n, D))
        traj_len = np.zeros(n, dtype=np.int32); status = np.zeros(n, dtype=np.int32)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        packed = _dp()
        self._check(self._lib.ltp_get_trajectory_host(self._h, n, _ptr(t), _ptr(f[0]), modb.ctypes.data_as(C.POINTER(C.c_byte)),
                                                      _ptr(f[1]), _ptr(f[2]), _ptr(f[3]), _ptr(f[4]),
                                                      traj_len.ctypes.data_as(C.POINTER(C.c_int)), status.ctypes.data_as(C.POINTER(C.c_int)),
                                                      offsets.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(packed)))
        total = int(offsets[n])
        out = np.ctypeslib.as_array(packed, shape=(max(total, 1),))[:total].copy()
        self._lib.ltp_free_host(packed)
        return dict(traj_len=traj_len, status=status, offsets=offsets, packed=out)

    @staticmethod
    def planBatchSharded(planners, q_goal, q_0, v_0, a_0, sample=True):
        """NEW: one process, several devices. `planners`: identically configured LongTermPlanner objects,
        normally one per device; shard g plans the contiguous query range shard_range(n, g, len(planners)) on its own
        device and host thread (ltp_plan_batch_multi); the result dict is that of ONE planBatchHost call over all queries."""
        lib = planners[0]._lib
        D = planners[0].dof
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, D)) for x in (q_goal, q_0, v_0, a_0)]
        n = ins[0].shape[0]
        r = dict(t_opt=np.zeros((n, D, 7)), t_scaled=np.zeros((n, D, 7)), dir=np.zeros((n, D)), v_drive=np.zeros((n, D)),
                 mod=np.zeros((n, D), dtype=np.int8), t_required=np.zeros(n), slowest=np.zeros(n, dtype=np.int32),
                 traj_len=np.zeros(n, dtype=np.int32), status=np.zeros(n, dtype=np.int32))
        rec = _abi.Records(*[r[k].ctypes.data for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status")])
        offsets = np.zeros(n + 1, dtype=np.uint64)
        packed = _dp()
        handles = (C.c_void_p * len(planners))(*[pl._h for pl in planners])
        rc = lib.ltp_plan_batch_multi(handles, len(planners), n, *[_ptr(x) for x in ins], C.byref(rec),
                                      offsets.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(packed) if sample else None)
        planners[0]._check(rc)
        r["offsets"] = offsets
        if sample:
            total = int(offsets[n])
            r["packed"] = np.ctypeslib.as_array(packed, shape=(max(total, 1),))[:total].copy()
            lib.ltp_free_host(packed)
        return r

    # ---- device-resident shards, one process: k planners, per-shard torch tensors, no host copies ----
    @staticmethod
    def _shard_args(planners, batches, streams=None):
        handles = (C.c_void_p * len(planners))(*[pl._h for pl in planners])
        shards = (_abi.Shard * len(planners))()
        for g, (pl, b) in enumerate(zip(planners, batches)):
            shards[g].in_ = b.queries
            shards[g].out = b.c_records()
            shards[g].offsets = b.offsets.data_ptr()
            shards[g].stream = streams[g].cuda_stream if streams is not None else pl._stream().value
        return handles, shards

    @staticmethod
    def planSwitchTimesSharded(planners, shard_inputs, n, layout="query_major", batches=None, end_limit=False, streams=None):
        """NEW: ltp_plan_switch_times_multi. shard_inputs[g] = (q_goal, q_0, v_0, a_0) CUDA tensors of shard g — the queries
        shard_range(n, g, k) of one batch of n — on planners[g]'s device. Returns the per-shard DeviceBatch list; the work
        is enqueued on streams[g] (default: each device's current stream), not waited for."""
        k = len(planners)
        if batches is None:
            batches = [None] * k
        out = []
        for g, (pl, ins) in enumerate(zip(planners, shard_inputs)):
            cnt, q = LongTermPlanner._queries(*ins, layout)
            b = batches[g]
            if b is None or b.n != cnt or b.dof != pl.dof:
                b = DeviceBatch(cnt, pl.dof, ins[1].device)
            b.queries, b.inputs = q, ins
            out.append(b)
        handles, shards = LongTermPlanner._shard_args(planners, out, streams)
        planners[0]._check(planners[0]._lib.ltp_plan_switch_times_multi(handles, k, int(n), shards, 1 if end_limit else 0))
        return out

    @staticmethod
    def envelopeSharded(planners, batches, n, window, n_windows, outs=None, streams=None):
        """NEW: ltp_envelope_multi over the shards planned by planSwitchTimesSharded; returns per-shard [count, dof, n_windows, 2]."""
        import torch
        k = len(planners)
        if outs is None:
            outs = [torch.empty((b.n, pl.dof, n_windows, 2), dtype=torch.float64, device=b.offsets.device) for pl, b in zip(planners, batches)]
        handles, shards = LongTermPlanner._shard_args(planners, batches, streams)
        envs = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
        planners[0]._check(planners[0]._lib.ltp_envelope_multi(handles, k, int(n), shards, int(window), int(n_windows), envs))
        return outs

    @staticmethod
    def stateAtSharded(planners, batches, n, sample_index, streams=None, outs=None):
        """NEW: ltp_state_at_multi; sample_index: int, or a list of per-shard int32 CUDA tensors. Returns per shard [q, v, a]
        laid out like the shard's queries (outs: reuse these tensors — with side streams, buffers that are allocated once
        are the safe choice: the caching allocator does not know about work pending on another stream)."""
        import torch
        k = len(planners)
        if outs is None:
            outs = [[torch.empty_like(b.inputs[1]) for _ in range(3)] for b in batches]
        handles, shards = LongTermPlanner._shard_args(planners, batches, streams)
        per = None if isinstance(sample_index, int) else (C.c_void_p * k)(*[x.data_ptr() for x in sample_index])
        ptrs = [(C.c_void_p * k)(*[o[i].data_ptr() for o in outs]) for i in range(3)]
        planners[0]._check(planners[0]._lib.ltp_state_at_multi(handles, k, int(n), shards, per, sample_index if per is None else 0, *ptrs))
        return outs

    @staticmethod
    def synchronizeSharded(planners, batches, streams=None):
        handles, shards = LongTermPlanner._shard_args(planners, batches, streams)
        planners[0]._check(planners[0]._lib.ltp_synchronize_multi(handles, len(planners), shards))

    @staticmethod
    def planEnvelopeSharded(planners, q_goal, q_0, v_0, a_0, window, n_windows):
        """NEW: planEnvelopeHost over several planners / devices from one process (ltp_plan_envelope_multi_host)."""
        lib = planners[0]._lib
        D = planners[0].dof
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, D)) for x in (q_goal, q_0, v_0, a_0)]
        n = ins[0].shape[0]
        r = dict(t_opt=np.zeros((n, D, 7)), t_scaled=np.zeros((n, D, 7)), dir=np.zeros((n, D)), v_drive=np.zeros((n, D)),
                 mod=np.zeros((n, D), dtype=np.int8), t_required=np.zeros(n), slowest=np.zeros(n, dtype=np.int32),
                 traj_len=np.zeros(n, dtype=np.int32), status=np.zeros(n, dtype=np.int32))
        rec = _abi.Records(*[r[k].ctypes.data for k in ("t_opt", "t_scaled", "dir", "v_drive", "mod", "t_required", "slowest", "traj_len", "status")])
        env = np.zeros((n, D, int(n_windows), 2))
        handles = (C.c_void_p * len(planners))(*[pl._h for pl in planners])
        planners[0]._check(lib.ltp_plan_envelope_multi_host(handles, len(planners), n, *[_ptr(x) for x in ins], int(window), int(n_windows),
                                                           C.byref(rec), _ptr(env)))
        return r, env

    def lastSamplerKernel(self):
        """Name of the kernel that wrote the rows / envelopes of this handle's latest sampleBatch / envelopeBatch call."""
        return (self._lib.ltp_last_sampler_kernel(self._h) or b"").decode()

    def reserveTables(self, n):
        self._check(self._lib.ltp_reserve_tables(self._h, int(n)))

    # ---- batched device calls (torch CUDA tensors, asynchronous on torch's current stream) ----
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _queries(q_goal, q_0, v_0, a_0, layout):
        n, D = (q_0.shape if layout == "query_major" else q_0.shape[::-1])
        sq, sj = (D, 1) if layout == "query_major" else (1, n)
        for x in (q_goal, q_0, v_0, a_0):
            assert x.is_cuda and x.is_contiguous() and x.dtype.is_floating_point and x.element_size() == 8
        return n, _abi.Queries(q_goal.data_ptr(), q_0.data_ptr(), v_0.data_ptr(), a_0.data_ptr(), sq, sj)

    def generateQueries(self, n, seed=12345, first_query=0, layout="query_major"):
        """Synthetic batch on the device (ltp_generate_queries_batch); returns q_goal, q_0, v_0, a_0."""
        import torch
        D = self.dof
        shape = (n, D) if layout == "query_major" else (D, n)
        sq, sj = (D, 1) if layout == "query_major" else (1, n)
        out = [torch.empty(shape, dtype=torch.float64, device=f"cuda:{self.device}") for _ in range(4)]
        self._check(self._lib.ltp_generate_queries_batch(self._h, n, seed, first_query, *[x.data_ptr() for x in out], sq, sj, self._stream()))
        return out

    def planSwitchTimesBatch(self, q_goal, q_0, v_0, a_0, layout="query_major", batch: Optional[DeviceBatch] = None,
                             end_limit=False):
        """Stages 1-3 + traj_len + packed offsets for a device batch (ltp_plan_switch_times_batch). end_limit=True also
        runs planTrajectory's end-limit check (cc:59-61) without sampling (ltp_end_limit_batch), so that status == 0 is
        exactly planTrajectory's bool; sampleBatch / envelopeBatch apply that check themselves."""
        n, q = self._queries(q_goal, q_0, v_0, a_0, layout)
        if batch is None or batch.n != n or batch.dof != self.dof:
            batch = DeviceBatch(n, self.dof, q_0.device)
        batch.queries = q
        batch.inputs = (q_goal, q_0, v_0, a_0)   # the sampler reads q_0/v_0/a_0 again: keep the tensors alive with the batch
        rec = batch.c_records()
        self._check(self._lib.ltp_plan_switch_times_batch(self._h, n, C.byref(q), C.byref(rec), batch.offsets.data_ptr(), self._stream()))
        if end_limit:
            self.endLimit(batch, 0, n)
        return batch

    def endLimit(self, batch: DeviceBatch, first, count):
        """planTrajectory's end-limit check (cc:59-61) for plans [first, first+count) without sampling (ltp_end_limit_batch)."""
        rec = batch.c_records()
        self._check(self._lib.ltp_end_limit_batch(self._h, first, count, C.byref(batch.queries), C.byref(rec), self._stream()))

    def sampleBatch(self, batch: DeviceBatch, first, count, out, streaming=True, dry=False, spread=0, tables=None, walk=None, auto_waves=None, verdict=True):
        """getTrajectory for plans [first, first+count) into the float64 CUDA tensor `out` (ltp_sample_batch).
        dry=True is a diagnostic: same stores, no arithmetic (ceiling of the store pattern). tables: None = automatic, True / False =
        force the table pass / the fused build; walk: None = automatic, True / False = force / forbid k_sample_walk_* (run tables built
        inside the sampler's block: by itself for capped, float32 and sparse rows and in MATLAB semantics); auto_waves=False keeps
        the walk kernel's builder / streaming-wave form also for caps of at most 32 samples (flag bit 7; by itself: k_sample_walk_auto_*);
        verdict=False (flag bit 4): capped rows without the end-limit verdict — the walk kernels stop at the cap, STATUS_END_LIMIT is then
        not formed by this call (same rows)."""
        import torch
        rec = batch.c_records()
        fn = self._lib.ltp_sample_batch_f32 if out.dtype == torch.float32 else self._lib.ltp_sample_batch   # float32 tile -> float rows
        assert out.dtype in (torch.float32, torch.float64)
        self._check(fn(self._h, first, count, C.byref(batch.queries), C.byref(rec), batch.offsets.data_ptr(),
                       out.data_ptr(), out.numel(), (1 if streaming else 0) | (2 if dry else 0) | (int(spread) << 8)
                       | (0 if tables is None else (4 if tables else 8)) | (0 if walk is None else (64 if walk else 32)) | (128 if auto_waves is False else 0) | (0 if verdict else 16), self._stream()))

    SAMPLERS = {"auto": 0, "fused": 1, "walk": 2, "walk_streaming": 3, "table": 4}

    def sampleBatchEx(self, batch: DeviceBatch, first, count, out, sampler="auto", nontemporal=True, verdict=True, interleave=0, dry=False):
        """getTrajectory for plans [first, first+count) through ltp_sample_batch_ex: the policy as the named fields of ltp_sample_opts
        (sampler: "auto" | "fused" | "walk" | "walk_streaming" | "table"). The same rows whatever the fields say."""
        import torch
        from ._abi import SampleOpts
        assert out.dtype in (torch.float32, torch.float64)
        o = SampleOpts(C.sizeof(SampleOpts), 1 if out.dtype == torch.float32 else 0, 0 if nontemporal else 1, self.SAMPLERS[sampler],
                       0 if verdict else 1, int(interleave), 1 if dry else 0)
        rec = batch.c_records()
        self._check(self._lib.ltp_sample_batch_ex(self._h, first, count, C.byref(batch.queries), C.byref(rec), batch.offsets.data_ptr(),
                                                  out.data_ptr(), out.numel(), C.addressof(o), self._stream()))

    def envelopeBatch(self, batch: DeviceBatch, first, count, window, n_windows, out=None):
        """NEW (on-device consumer): [count, dof, n_windows, 2] = min / max of q over windows of `window`
        samples of plans [first, first+count), without storing the trajectories (ltp_envelope_batch)."""
        import torch
        if out is None:
            out = torch.empty((count, self.dof, n_windows, 2), dtype=torch.float64, device=batch.offsets.device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float64 and out.numel() >= count * self.dof * n_windows * 2
        rec = batch.c_records()
        self._check(self._lib.ltp_envelope_batch(self._h, first, count, C.byref(batch.queries), C.byref(rec), int(window),
                                                 int(n_windows), out.data_ptr(), self._stream()))
        return out

    def buildRunTables(self, batch: DeviceBatch, first, count, out=None):
        """NEW (the consumer HOOK): the packed run tables of plans [first, first+count) in a caller buffer
        (ltp_build_tables_batch; format and device functions: include/ltp_run_tables.hpp). Returns an int64 CUDA tensor of
        ltp_run_tables_bytes / 8 words; lane (plan - first) * dof + joint."""
        import torch
        words = int(self._lib.ltp_run_tables_bytes(self._h, count)) // 8
        if out is None:
            out = torch.empty(max(words, 2), dtype=torch.int64, device=batch.offsets.device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.int64      # (the size is checked by the library)
        rec = batch.c_records()
        self._check(self._lib.ltp_build_tables_batch(self._h, first, count, C.byref(batch.queries), C.byref(rec), out.data_ptr(),
                                                     out.numel() * 8, self._stream()))
        return out

    def replanStates(self, batch: DeviceBatch, first, count, tile, sample_index, layout="query_major"):
        """NEW: start states (q_0, v_0, a_0) of the next plans = sample k of the trajectories that
        sampleBatch(batch, first, count, tile) wrote. sample_index: int or int32 CUDA tensor [count]."""
        import torch
        D = self.dof
        shape = (count, D) if layout == "query_major" else (D, count)
        sq, sj = (D, 1) if layout == "query_major" else (1, count)
        out = [torch.empty(shape, dtype=torch.float64, device=tile.device) for _ in range(3)]
        rec = batch.c_records()
        per_plan = None if isinstance(sample_index, int) else sample_index
        fn = self._lib.ltp_replan_states_f32_batch if tile.dtype == torch.float32 else self._lib.ltp_replan_states_batch
        self._check(fn(self._h, first, count, C.byref(batch.queries), C.byref(rec), batch.offsets.data_ptr(),
                       tile.data_ptr(), tile.numel(), per_plan.data_ptr() if per_plan is not None else None,
                       sample_index if per_plan is None else 0, *[x.data_ptr() for x in out], sq, sj, self._stream()))
        return out

    def stateAt(self, batch: DeviceBatch, first, count, sample_index, layout="query_major"):
        """NEW: (q, v, a) at trajectory sample k of plans [first, first+count) straight from the records
        (ltp_state_at_batch) — no sampled rows needed. sample_index: int or int32 CUDA tensor [count]."""
        import torch
        D = self.dof
        shape = (count, D) if layout == "query_major" else (D, count)
        sq, sj = (D, 1) if layout == "query_major" else (1, count)
        out = [torch.empty(shape, dtype=torch.float64, device=batch.offsets.device) for _ in range(3)]
        rec = batch.c_records()
        per_plan = None if isinstance(sample_index, int) else sample_index
        self._check(self._lib.ltp_state_at_batch(self._h, first, count, C.byref(batch.queries), C.byref(rec),
                                                 per_plan.data_ptr() if per_plan is not None else None,
                                                 sample_index if per_plan is None else 0, *[x.data_ptr() for x in out], sq, sj, self._stream()))
        return out

    def roots(self, poly, dtype=np.float64):
        """roots<T>() of the reference's roots.h:22-34 on the device: all eigenvalues of the companion matrix of each
        polynomial (rows of `poly`, highest coefficient first), as a complex array in Eigen's output order."""
        dt = np.dtype(dtype)
        poly = np.ascontiguousarray(np.atleast_2d(np.asarray(poly, dtype=dt)))
        n, deg = poly.shape[0], poly.shape[1] - 1
        re = np.zeros((n, deg), dtype=dt); im = np.zeros((n, deg), dtype=dt)
        if dt == np.float32:
            fp = C.POINTER(C.c_float)
            self._check(self._lib.ltp_roots_f32_host(self._h, n, deg, poly.ctypes.data_as(fp), re.ctypes.data_as(fp), im.ctypes.data_as(fp)))
        else:
            self._check(self._lib.ltp_roots_f64_host(self._h, n, deg, _ptr(poly), _ptr(re), _ptr(im)))
        return re + 1j * im

    # ---- diagnostics for the parity tests ----
    def debugMathProbe(self, x, y):
        x, y = _vec(x), _vec(y)
        out = np.zeros((x.size, 8))
        self._check(self._lib.ltp_debug_math_probe_host(self._h, x.size, _ptr(x), _ptr(y), _ptr(out)))
        return out

    @staticmethod
    def powRuleMatchingHostLibm(probes=0):
        """NEW: ("libm" | "exact" | None, mismatches vs the libm rule, mismatches vs the exact rule) — which pow rule has the bits of the
        C library this process runs on, i.e. of a reference built on this host (ltp_hip.h ltp_host_libm_pow_rule; host only)."""
        import ctypes as C
        from . import _abi
        a, b = C.c_longlong(), C.c_longlong()
        r = _abi.lib().ltp_host_libm_pow_rule(int(probes), C.byref(a), C.byref(b))
        return ({1: "libm", 0: "exact"}.get(r), a.value, b.value)

    def debugLibmPow(self, x, y):
        """pow(x, y) elementwise by the device's restated glibc pow (the arithmetic of setPowRule("libm"))."""
        x, y = _vec(x), _vec(y)
        if y.size != x.size:
            x, y = (np.ascontiguousarray(a, dtype=np.float64) for a in np.broadcast_arrays(x, y))     # the C call reads x.size of both
        out = np.zeros(x.size)
        self._check(self._lib.ltp_debug_libm_pow_host(self._h, x.size, _ptr(x), _ptr(y), _ptr(out)))
        return out

    def debugRootsProbe(self, degree, coef):
        coef = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(-1, 7))
        out = np.zeros(coef.shape[0])
        self._check(self._lib.ltp_debug_roots_probe_host(self._h, coef.shape[0], int(degree), _ptr(coef), _ptr(out)))
        return out
