"""What one MATLAB-semantics eigen-solve costs on the device: k_roots_matlab (the kernel behind LongTermPlanner.matlabRoots) on the
polynomials a 100 k panda batch really solves (tools/exp/polys100k.npy: rows [degree, c0..c6, root] logged by the oracle), alone and
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

rows = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "polys100k.npy"))
D, lim = amd.limit_set("panda")
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
