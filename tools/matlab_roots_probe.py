"""What one MATLAB-semantics eigen-solve costs on the device: k_roots_matlab (the kernel behind LongTermPlanner.matlabRoots) on the
polynomials a 100 k panda batch really solves (logged by the oracle while it plans the same batch), alone and
in company. Run under rocprofv3 --kernel-trace (durations) or --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES (instructions):
the launches appear in this order — per degree 4, 5, 6: [worst polynomial x 64 lanes = one wave] [worst polynomial x 1 lane]
[every polynomial of the degree]."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import longtermplanner_amd as amd
import oracle

D, lim = amd.limit_set("panda")
# the polynomials of bench.py's 100 k panda batch (seed 12345), logged by the oracle's MATLAB-semantics twin: rows [degree, c0..c6, root]
qg, q0, v0, a0 = amd.generate_queries(100000, lim, seed=12345)
orc = oracle.Oracle(D, 0.001, semantics="matlab", **lim)
_, rows = oracle.poly_log(lambda: orc.plan_batch(qg, q0, v0, a0, sample=False), cap=3000000)
ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
lib = oracle.lib()
ltp.matlabRoots(np.array([[1.0, -3.0, 2.0]]))      # warm-up launch
for deg in (4, 5, 6):
    polys = np.ascontiguousarray(np.array([r[1:2 + deg] for r in rows if int(r[0]) == deg]))
    sweeps = []
    for c in polys:
        oracle.matlab_roots(c)
        m = C.c_int(0)
        sweeps.append(lib.ltpm_debug_iters(C.byref(m)))
    sweeps = np.array(sweeps)
    worst = polys[int(np.argmax(sweeps))]
    print(f"degree {deg}: {len(polys)} polynomials, sweeps mean {sweeps.mean():.1f} max {sweeps.max()}", flush=True)
    ltp.matlabRoots(np.tile(worst, (64, 1)))
    ltp.matlabRoots(worst[None, :])
    ltp.matlabRoots(polys)
