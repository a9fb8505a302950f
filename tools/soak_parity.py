"""Large-sample parity soak: switching-time records of the HIP path vs the CPU oracle on millions of queries.
usage: python tools/soak_parity.py [--matlab] [n] [seed ...]      (--matlab: the MATLAB-semantics mode against its oracle twin)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import longtermplanner_amd as amd
import oracle

argv = [a for a in sys.argv[1:] if a != "--matlab"]
matlab = "--matlab" in sys.argv[1:]
n = int(argv[0]) if len(argv) > 0 else 2_000_000
seeds = [int(x) for x in argv[1:]] or [777]
for seed in seeds:
  for name in ("panda", "ref", "ref30"):
      D, lim = amd.limit_set(name)
      nn = n if D == 7 else n // 5
      ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
      orc = oracle.Oracle(D, 0.001, semantics="matlab" if matlab else "cpp", **lim)
      if matlab:
          ltp.setSemantics("matlab")
      qg, q0, v0, a0 = amd.generate_queries(nn, lim, seed=seed)
      t0 = time.time()
      r = ltp.planBatchHost(qg, q0, v0, a0, sample=False)
      t1 = time.time()
      parts = 16
      with ThreadPoolExecutor(parts) as ex:
          outs = list(ex.map(lambda i: orc.plan_batch(qg[i::parts], q0[i::parts], v0[i::parts], a0[i::parts], sample=False), range(parts)))
      t2 = time.time()
      worst, flips, over, bits = 0.0, 0, 0, 0
      for i, o in enumerate(outs):
          sl = slice(i, None, parts)
          ok = o["status"] != 0
          flips += int(np.sum(((r["status"][sl] & (7 | 128)) == 0) != ok))
          if matlab:
              flips += int(np.sum(((r["status"][sl] & 128) != 0) != ((o["matlab_flags"] & 2) != 0)))
              flips += int(np.sum(((r["status"][sl] & 256) != 0) != ((o["matlab_flags"] & 1) != 0)))
          for k in ("slowest", "mod", "traj_len"):
              flips += int(np.sum(r[k][sl][ok] != o[k][ok]))
          flips += int(np.sum(r["dir"][sl][ok] != o["dir"][ok]))
          for k in ("t_opt", "t_scaled", "v_drive", "t_required"):
              d = np.abs(r[k][sl][ok] - o[k][ok]); d = d[np.isfinite(d)]
              worst = max(worst, float(d.max())); over += int(np.sum(d > 1e-9))
              x, y = np.ascontiguousarray(r[k][sl][ok]), np.ascontiguousarray(o[k][ok])
              bits += int(np.sum((x.view(np.uint64) != y.view(np.uint64)) & ~(np.isnan(x) & np.isnan(y))))   # entries whose BITS differ (default pow rule: none)
      print(("MATLAB semantics " if matlab else "") + f"seed {seed} {name:6s} {nn:8d} queries ({nn * D} joint lanes): integer mismatches {flips}, values beyond 1e-9: {over}, worst |d| {worst:.3e}, entries with other bits {bits}"
            f"  [gpu+copies {t1 - t0:.1f} s, oracle x16 threads {t2 - t1:.1f} s]", flush=True)
