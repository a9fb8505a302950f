"""Diagnostic: where one k_envelope work item spends its time (100 MHz wall-clock stamps per phase)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longtermplanner_amd import LongTermPlanner, limit_set
name = sys.argv[2] if len(sys.argv) > 2 else "panda"
dof, lim = limit_set(name)
ltp = LongTermPlanner(dof, 0.001, device=0, **lim)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
W, K = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "64:32").split(":"))
qg, q0, v0, a0 = ltp.generateQueries(n)
b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
groups = (dof + 7) // 8
env = ltp.envelopeBatch(b, 0, n, W, K)
torch.cuda.synchronize()
stamps = torch.zeros(16 * n * groups, dtype=torch.int64, device="cuda")
ltp._lib.ltp_debug_set_sample_stamps(ltp._h, stamps.data_ptr())
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ltp.envelopeBatch(b, 0, n, W, K, out=env); e1.record()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 10] > 0]
    names = ["draw item (atomic + barrier)", "traj_len load", "(1) record loads, floor/ceil", "(2) jerks, corrections, cut candidates",
             "(3a) unique", "(3b) rank", "(4) run mode / jerk", "(5) serial state walk", "(6) coefficients", "window reduction"]
    d = np.diff(st[:, :11].astype(np.float64), axis=1) / 100.0
    print(f"rep {rep}: launch {e0.elapsed_time(e1):.2f} ms for {n} plans; mean us per phase:")
    for i, nm in enumerate(names):
        print(f"   {nm:42s} {d[:, i].mean():7.2f}")
    print(f"   {'whole item':42s} {((st[:, 10] - st[:, 0]) / 100.0).mean():7.2f}")
