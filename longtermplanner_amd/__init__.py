"""longtermplanner_amd — MI355X (gfx950) batched jerk-limited trajectory planner.

Drop-in for the hot path of yannickBurkhardt/LongTermPlanner (``LongTermPlanner::planTrajectory`` and the
member functions under it) with a batched overload: hand-written HIP kernels behind a C ABI
(``include/ltp_hip.h`` -> ``longtermplanner_amd/libltp_hip.so``). No CPU fallback exists.
"""
from ._abi import (LtpError, SEMANTICS_CPP, SEMANTICS_MATLAB, STATUS_END_LIMIT, STATUS_GOAL_OUTSIDE, STATUS_INVALID_INPUT,  # noqa: F401
                   STATUS_MATLAB_COMPLEX, STATUS_MATLAB_ERROR, STATUS_NO_SLOWEST, STATUS_NONFINITE, STATUS_OPT_FAILED, STATUS_OVERFLOW)
from .planner import DeviceBatch, LongTermPlanner, Trajectory, unpack_trajectory  # noqa: F401
from .synthetic import generate_queries, limit_set  # noqa: F401
