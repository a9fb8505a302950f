"""ctypes binding of libltp_hip.so (C ABI declared in include/ltp_hip.h).

The library is built in-tree by ``longtermplanner_amd/csrc/Makefile`` (hipcc, gfx950) and must be
present: there is no CPU or PyTorch fallback behind this module. A missing library, or a machine
without a HIP device, raises.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libltp_hip.so")
CSRC = os.path.join(_PKG, "csrc")

LTP_OK = 0
STATUS_INVALID_INPUT = 1
STATUS_OPT_FAILED = 2
STATUS_NO_SLOWEST = 4
STATUS_END_LIMIT = 8
STATUS_NONFINITE = 16
STATUS_OVERFLOW = 32
STATUS_GOAL_OUTSIDE = 64
STATUS_MATLAB_ERROR = 128
STATUS_MATLAB_COMPLEX = 256
SEMANTICS_CPP = 0
SEMANTICS_MATLAB = 1
POW_EXACT = 0
POW_LIBM = 1

ERROR_NAMES = {1: "LTP_ERR_INVALID_ARGUMENT", 2: "LTP_ERR_NO_DEVICE", 3: "LTP_ERR_OUT_OF_MEMORY", 4: "LTP_ERR_HIP"}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_byte)
_up = C.POINTER(C.c_ulonglong)


class LtpError(RuntimeError):
    def __init__(self, code, text=""):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {text}")
        self.code = code


class SampleOpts(C.Structure):
    """ltp_sample_opts (include/ltp_hip.h): the sampler's policy as named fields, size-versioned."""
    _fields_ = [("size", C.c_uint), ("format", C.c_int), ("stores", C.c_int), ("sampler", C.c_int), ("verdict", C.c_int),
                ("interleave", C.c_int), ("dry_run", C.c_int)]


class Queries(C.Structure):
    _fields_ = [("q_goal", C.c_void_p), ("q_0", C.c_void_p), ("v_0", C.c_void_p), ("a_0", C.c_void_p),
                ("query_stride", C.c_longlong), ("joint_stride", C.c_longlong)]


class Records(C.Structure):
    _fields_ = [("t_opt", C.c_void_p), ("t_scaled", C.c_void_p), ("dir", C.c_void_p), ("v_drive", C.c_void_p),
                ("mod", C.c_void_p), ("t_required", C.c_void_p), ("slowest", C.c_void_p), ("traj_len", C.c_void_p),
                ("status", C.c_void_p)]


class Shard(C.Structure):
    """ltp_shard: one device-resident shard of a *_multi call (include/ltp_hip.h)."""
    _fields_ = [("in_", Queries), ("out", Records), ("offsets", C.c_void_p), ("stream", C.c_void_p)]


def build_inputs():
    """Every file libltp_hip.so is made from, as the Makefile itself lists them (`make print-deps`: the kernels, the C ABI sources
    and ALL their headers, the public device header include/ltp_run_tables.hpp among them) — one list, kept in one place."""
    out = subprocess.check_output(["make", "-C", CSRC, "-s", "--no-print-directory", "print-deps"], text=True)
    return [os.path.normpath(os.path.join(CSRC, f)) for f in out.split()]


def stale():
    """True if libltp_hip.so is missing or older than one of its inputs. (Object files do not travel to the GPU box, so `make -q`
    cannot be asked there: it would call an intact library out of date.)"""
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > built for f in build_inputs())


def build(force=False):
    """Compile libltp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or stale():
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4", "all"])
        os.utime(LIB_PATH)          # make may have found the objects current (an input was only touched): the library is current too
    return LIB_PATH


_SIGNATURES = {
    "ltp_create": (C.c_int, [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, C.c_int, C.POINTER(C.c_void_p)]),
    "ltp_destroy": (None, [C.c_void_p]),
    "ltp_set_limits": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "ltp_set_sample_time": (C.c_int, [C.c_void_p, C.c_double]),
    "ltp_set_dof": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_dof": (C.c_int, [C.c_void_p]),
    "ltp_get_sample_time": (C.c_double, [C.c_void_p]),
    "ltp_last_error": (C.c_char_p, [C.c_void_p]),
    "ltp_row_stride": (C.c_int, [C.c_int]),
    "ltp_set_max_samples": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_max_samples": (C.c_int, [C.c_void_p]),
    "ltp_stored_samples": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_set_sample_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_sample_stride": (C.c_int, [C.c_void_p]),
    "ltp_run_tables_bytes": (C.c_ulonglong, [C.c_void_p, C.c_longlong]),
    "ltp_build_tables_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                        C.c_ulonglong, C.c_void_p]),
    "ltp_envelope_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "ltp_debug_set_sample_blocks": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_debug_get_sample_blocks": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_reserve_batch": (C.c_int, [C.c_void_p, C.c_longlong]),
    "ltp_reserve_tables": (C.c_int, [C.c_void_p, C.c_longlong]),
    "ltp_last_sampler_kernel": (C.c_char_p, [C.c_void_p]),
    "ltp_plan_switch_times_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, C.POINTER(Shard), C.c_int]),
    "ltp_envelope_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, C.POINTER(Shard), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ltp_state_at_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, C.POINTER(Shard), C.POINTER(C.c_void_p), C.c_int,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ltp_synchronize_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Shard)]),
    "ltp_plan_envelope_multi_host": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                               C.POINTER(Records), _dp]),
    "ltp_plan_envelope_host": (C.c_int, [C.c_void_p, C.c_longlong, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.POINTER(Records), _dp]),
    "ltp_state_at_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "ltp_set_semantics": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_semantics": (C.c_int, [C.c_void_p]),
    "ltp_set_envelope_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_envelope_mode": (C.c_int, [C.c_void_p]),
    "ltp_set_pow_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_pow_rule": (C.c_int, [C.c_void_p]),
    "ltp_host_libm_pow_rule": (C.c_int, [C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "ltp_debug_libm_pow_host": (C.c_int, [C.c_void_p, C.c_longlong, _dp, _dp, _dp]),
    "ltp_debug_last_matlab_flags": (C.c_int, [C.c_void_p]),
    "ltp_debug_roots_matlab_host": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, _dp, _dp, _dp, _ip, _ip]),
    "ltp_set_goal_check": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_goal_check": (C.c_int, [C.c_void_p]),
    "ltp_set_table_pass": (C.c_int, [C.c_void_p, C.c_int]),
    "ltp_get_table_pass": (C.c_int, [C.c_void_p]),
    "ltp_set_table_workspace": (C.c_int, [C.c_void_p, C.c_ulonglong]),
    "ltp_replan_states_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                          C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                          C.c_longlong, C.c_void_p]),
    "ltp_plan_switch_times_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p, C.c_void_p]),
    "ltp_sample_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                   C.c_void_p, C.c_ulonglong, C.c_int, C.c_void_p]),
    "ltp_sample_batch_ex": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                      C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "ltp_sample_batch_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                       C.c_void_p, C.c_ulonglong, C.c_int, C.c_void_p]),
    "ltp_replan_states_f32_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p,
                                              C.c_void_p, C.c_ulonglong, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                              C.c_longlong, C.c_void_p]),
    "ltp_end_limit_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.POINTER(Queries), C.POINTER(Records), C.c_void_p]),
    "ltp_shard_range": (None, [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "ltp_plan_batch_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_longlong, _dp, _dp, _dp, _dp, C.POINTER(Records), _up,
                                       C.POINTER(_dp)]),
    "ltp_generate_queries_batch": (C.c_int, [C.c_void_p, C.c_longlong, C.c_ulonglong, C.c_longlong, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "ltp_plan_batch_host": (C.c_int, [C.c_void_p, C.c_longlong, _dp, _dp, _dp, _dp, C.POINTER(Records), _up, C.POINTER(_dp)]),
    "ltp_get_trajectory_host": (C.c_int, [C.c_void_p, C.c_longlong, _dp, _dp, _bp, _dp, _dp, _dp, _dp, _ip, _ip, _up, C.POINTER(_dp)]),
    "ltp_free_host": (None, [C.c_void_p]),
    "ltp_check_inputs_host": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _ip]),
    "ltp_opt_braking_host": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp]),
    "ltp_opt_switch_times_host": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                            _dp, _dp, C.c_char_p, _ip]),
    "ltp_time_scaling_host": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_double, _dp, _dp, C.c_char_p, _ip, _ip]),
    "ltp_roots_f64_host": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, _dp, _dp, _dp]),
    "ltp_roots_f32_host": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ltp_debug_set_sample_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ltp_debug_math_probe_host": (C.c_int, [C.c_void_p, C.c_longlong, _dp, _dp, _dp]),
    "ltp_debug_roots_probe_host": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, _dp, _dp]),
}

_lib = None


def lib():
    """Load libltp_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the planner has no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7. If torch is
        # going to be used in this process (streams, device tensors, torch.distributed) it has to be loaded
        # first so that libltp_hip.so binds to the same runtime; two runtimes cannot both open the GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)
