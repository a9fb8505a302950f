"""CPU oracle for the planTrajectory hot path — TEST INFRASTRUCTURE ONLY.

ctypes loader for ``oracle/libltp_oracle.so`` (built from ``ltp_oracle.c`` by
``oracle/Makefile``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``longtermplanner_amd/``, ``include/``) never does.

Each wrapper names the reference function it restates
(paths relative to /root/reference):
  Oracle.check_inputs      src/long_term_planner.cc:68-77
  Oracle.opt_braking       src/long_term_planner.cc:650-701
  Oracle.opt_switch_times  src/long_term_planner.cc:82-353
  Oracle.time_scaling      src/long_term_planner.cc:358-645
  Oracle.get_trajectory    src/long_term_planner.cc:706-841
  Oracle.plan_trajectory   src/long_term_planner.cc:7-63
  roots_f64 / roots_f32    include/long_term_planner/roots.h:22-34
  smallest_root            include/long_term_planner/roots.h:43-50
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libltp_oracle.so")
_EXACT_LIB_PATH = os.path.join(_HERE, "libltp_oracle_exactpow.so")   # diagnostic twin, see lib(exact_pow=True)


def build(force=False):
    """Compile the C restatement (gcc). Building the checker is not using it."""
    src_time = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ltp_oracle.c", "kat_grid.c", "companion_roots.inc", "matlab_roots.inc", "Makefile"))
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < src_time for p in (_LIB_PATH, _EXACT_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _LIB_PATH


class _Planner(C.Structure):
    _fields_ = [("dof", C.c_int), ("t_sample", C.c_double),
                ("q_min", C.POINTER(C.c_double)), ("q_max", C.POINTER(C.c_double)),
                ("v_max", C.POINTER(C.c_double)), ("a_max", C.POINTER(C.c_double)),
                ("j_max", C.POINTER(C.c_double)), ("semantics", C.c_int)]


_lib = None
_lib_exact = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_cp = C.POINTER(C.c_char)


def _load(path):
    if not os.path.exists(path):
        build()
    l = C.CDLL(path)
    l.ltpo_smallest_root.restype = C.c_double
    l.ltpo_plan_batch.restype = C.c_long
    l.ltpo_poly_log_end.restype = C.c_long
    l.ltpo_compare_dense.restype = C.c_longlong
    return l


def lib(exact_pow=False):
    """The checker library. exact_pow=True: the DIAGNOSTIC twin built with -DLTPO_EXACT_POW, whose pow(x, 3 | 4 | 6) is one
    rounding of the exact product and whose pow(x, 0.5) is sqrt — the device's rule (csrc/ltp_math.hpp) — instead of libm's
    pow. It exists to show that libm's pow is the ONLY source of last-bit differences between the device and the default
    oracle; the parity reference is always the default build."""
    global _lib, _lib_exact
    if exact_pow:
        if _lib_exact is None:
            _lib_exact = _load(_EXACT_LIB_PATH)
            assert _lib_exact.ltpo_exact_pow() == 1
        return _lib_exact
    if _lib is None:
        _lib = _load(_LIB_PATH)
        assert _lib.ltpo_exact_pow() == 0
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    if n is not None and a.size == 1 and n != 1:
        a = np.full(n, float(a[0]))
    return a


def libm_pow(x, y):
    """The host libm's pow, elementwise (numpy's own power may take a SIMD path with other last bits)."""
    x = _arr(x); y = _arr(y, x.size)
    out = np.empty(x.size)
    f = lib().ltpo_libm_pow
    f.restype = None
    f(C.c_long(x.size), _d(x), _d(y), _d(out))
    return out


def roots_f64(poly):
    p = _arr(poly)
    n = p.size - 1
    re = np.empty(n); im = np.empty(n)
    st = lib().ltpo_roots_f64(_d(p), C.c_int(n), _d(re), _d(im))
    return re, im, st


def roots_f32(poly):
    p = np.ascontiguousarray(np.asarray(poly, dtype=np.float32))
    n = p.size - 1
    re = np.empty(n, dtype=np.float32); im = np.empty(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    st = lib().ltpo_roots_f32(p.ctypes.data_as(fp), C.c_int(n), re.ctypes.data_as(fp), im.ctypes.data_as(fp))
    return re, im, st


def smallest_root(poly):
    p = _arr(poly)
    return float(lib().ltpo_smallest_root(_d(p), C.c_int(p.size - 1)))


class Oracle:
    """Mirror of the reference class (ctor args as long_term_planner.h:118-131)."""

    def __init__(self, dof, t_sample, q_min, q_max, v_max, a_max, j_max, semantics="cpp", exact_pow=False):
        """semantics: "cpp" = src/long_term_planner.cc (the parity reference), "matlab" = LTPlanner.m where it diverges
        (LTPlanner.m has no position limits: q_min / q_max are ignored then)."""
        self.dof = int(dof)
        self.t_sample = float(t_sample)
        self.semantics = {"cpp": 0, "matlab": 1}[semantics]
        self.exact_pow = bool(exact_pow)      # diagnostic twin (lib(exact_pow=True)); never the parity reference
        self._lib = lib(self.exact_pow)
        self.set_limits(q_min, q_max, v_max, a_max, j_max)

    def set_limits(self, q_min, q_max, v_max, a_max, j_max):
        D = self.dof
        self.q_min, self.q_max = _arr(q_min, D), _arr(q_max, D)
        self.v_max, self.a_max, self.j_max = _arr(v_max, D), _arr(a_max, D), _arr(j_max, D)
        self._sync()

    def set_sample_time(self, t_sample):
        self.t_sample = float(t_sample)
        self._sync()

    def _sync(self):
        self._p = _Planner(self.dof, self.t_sample, _d(self.q_min), _d(self.q_max), _d(self.v_max), _d(self.a_max), _d(self.j_max),
                           getattr(self, "semantics", 0))

    @property
    def _ref(self):
        return C.byref(self._p)

    def check_inputs(self, q_0, v_0, a_0):
        return bool(self._lib.ltpo_check_inputs(self._ref, _d(_arr(q_0)), _d(_arr(v_0)), _d(_arr(a_0))))

    def opt_braking(self, joint, v_0, a_0):
        q = C.c_double(); d = C.c_double(); t = np.zeros(7)
        self._lib.ltpo_opt_braking(self._ref, C.c_int(joint), C.c_double(v_0), C.c_double(a_0), C.byref(q), _d(t), C.byref(d))
        return q.value, t, d.value

    def opt_switch_times(self, joint, q_goal, q_0, v_0, a_0, v_drive, t_init=None):
        t = np.zeros(7) if t_init is None else np.array(t_init, dtype=np.float64)
        d = C.c_double(); m = C.c_char()
        ok = self._lib.ltpo_opt_switch_times(self._ref, C.c_int(joint), C.c_double(q_goal), C.c_double(q_0), C.c_double(v_0),
                                         C.c_double(a_0), C.c_double(v_drive), _d(t), C.byref(d), C.byref(m))
        return bool(ok), t, d.value, ord(m.value)

    def time_scaling(self, joint, q_goal, q_0, v_0, a_0, dir_, t_required):
        t = np.zeros(7); vd = C.c_double(); m = C.c_char(); case = C.c_int()
        ok = self._lib.ltpo_time_scaling_ex(self._ref, C.c_int(joint), C.c_double(q_goal), C.c_double(q_0), C.c_double(v_0),
                                        C.c_double(a_0), C.c_double(dir_), C.c_double(t_required), _d(t), C.byref(vd),
                                        C.byref(m), C.byref(case))
        return bool(ok), t, vd.value, ord(m.value), case.value

    def traj_len(self, t):
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(self.dof, 7))
        return int(self._lib.ltpo_traj_len(self._ref, _d(t)))

    def get_trajectory(self, t, dir_, mod, q_0, v_0, a_0, v_drive):
        """Returns (length, q, v, a, j) with arrays [dof][length]."""
        D = self.dof
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(D, 7))
        n = self.traj_len(t)
        out = [np.zeros((D, max(n, 0))) for _ in range(4)]
        modb = np.ascontiguousarray(np.asarray(mod, dtype=np.int8).reshape(D))
        if n > 0:
            self._lib.ltpo_get_trajectory(self._ref, _d(t), _d(_arr(dir_)), modb.ctypes.data_as(_cp), _d(_arr(q_0)), _d(_arr(v_0)),
                                      _d(_arr(a_0)), _d(_arr(v_drive)), C.c_int(n), _d(out[0]), _d(out[1]), _d(out[2]), _d(out[3]))
        return (n, *out)

    def plan_batch(self, q_goal, q_0, v_0, a_0, sample=False, first=0, count=None, want_records=True):
        """Stages 1-3 (+ sampler when sample=True; sample="flat": sampler into arrays allocated once and reused instead
        of the reference's per-plan allocation) for row-major [n][dof] queries."""
        D = self.dof
        qg = np.ascontiguousarray(np.asarray(q_goal, dtype=np.float64).reshape(-1, D))
        q0 = np.ascontiguousarray(np.asarray(q_0, dtype=np.float64).reshape(-1, D))
        v0 = np.ascontiguousarray(np.asarray(v_0, dtype=np.float64).reshape(-1, D))
        a0 = np.ascontiguousarray(np.asarray(a_0, dtype=np.float64).reshape(-1, D))
        n = qg.shape[0]
        count = n - first if count is None else count
        r = {}
        if want_records:
            r = dict(t_opt=np.zeros((n, D, 7)), t_scaled=np.zeros((n, D, 7)), dir=np.zeros((n, D)),
                     mod=np.zeros((n, D), dtype=np.int8), v_drive=np.zeros((n, D)), t_required=np.zeros(n),
                     slowest=np.zeros(n, dtype=np.int32), traj_len=np.zeros(n, dtype=np.int32),
                     status=np.zeros(n, dtype=np.int32), checksum=np.zeros(n))
        null_d = C.cast(None, _dp)

        def g(k, cast=_dp):
            return r[k].ctypes.data_as(cast) if want_records else C.cast(None, cast)
        n_ok = self._lib.ltpo_plan_batch(self._ref, C.c_long(first), C.c_long(count), _d(qg), _d(q0), _d(v0), _d(a0),
                                     C.c_int(2 if sample == "flat" else (1 if sample else 0)), g("t_opt"), g("t_scaled"), g("dir"), g("mod", _cp),
                                     g("v_drive"), g("t_required"), g("slowest", _ip), g("traj_len", _ip),
                                     g("status", _ip), g("checksum"))
        del null_d
        r["n_ok"] = int(n_ok)
        if want_records and self.semantics == 1:
            # MATLAB semantics: bits 4 / 5 of the C status word = LTPlanner.m would have carried a complex value / raised an error
            r["matlab_flags"] = r["status"] >> 4
            r["status"] = r["status"] & 15
        return r

    def plan_trajectory(self, q_goal, q_0, v_0, a_0):
        """One full planTrajectory. Returns dict with status (0/1/2) and dense [dof][len] arrays."""
        D = self.dof
        rec = self.plan_batch(q_goal, q_0, v_0, a_0, sample=False)
        out = {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in rec.items()}
        if out["status"] == 0 or out["traj_len"] <= 0:
            out.update(status=0, length=0)
            return out
        n, q, v, a, j = self.get_trajectory(out["t_scaled"], out["dir"], out["mod"], q_0, v_0, a_0, out["v_drive"])
        st = 1
        for i in range(D):
            if q[i, n - 1] < self.q_min[i] or q[i, n - 1] > self.q_max[i]:
                st = 2
                break
        out.update(status=st, length=n, q=q, v=v, a=a, j=j)
        return out


def matlab_roots(poly):
    """MATLAB's roots() as LTPlanner.m sees it (oracle/matlab_roots.inc): complex array in MATLAB's output order."""
    p = _arr(poly)
    n = p.size - 1
    re = np.empty(n); im = np.empty(n); nr = C.c_int()
    st = lib().ltpm_roots(_d(p), C.c_int(n), _d(re), _d(im), C.byref(nr))
    return (re + 1j * im)[:nr.value], st


def poly_log(fn, cap=200000):
    """Run fn() while recording every polynomial solved: rows [degree, p0..p6, root]."""
    buf = np.zeros((cap, 9))
    lib().ltpo_poly_log_begin(_d(buf), C.c_long(cap))
    try:
        res = fn()
    finally:
        n = lib().ltpo_poly_log_end()
    return res, buf[:n].copy()
