import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import longtermplanner_amd as amd
D, lim = amd.limit_set("panda")
N = 200_000
ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
qg, q0, v0, a0 = ltp.generateQueries(N, seed=4711)
CAP = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ltp.setMaxSamples(CAP)
s0, s1, s2 = q0, v0, a0
def frac_wide(b, cap):
    tab = ltp.buildRunTables(b, 0, N)
    torch.cuda.synchronize()
    t = tab.cpu().numpy().view(np.uint64)
    lanes = N * D
    tiles = (lanes + 63) // 64
    t = t[:tiles * 114 * 64].reshape(tiles, 57, 64, 2)          # [tile][pair][lane][word in pair]
    w = t.transpose(0, 2, 1, 3).reshape(tiles * 64, 114)[:lanes]  # [lane][word]
    nseg = (w[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
    starts = w[:, 1:12].copy().view(np.int32).reshape(lanes, 22)
    idx = np.arange(22)[None, :]
    inside = ((idx < nseg[:, None]) & (starts < cap)).sum(axis=1)
    per_plan = inside.reshape(N, D).max(axis=1)
    return float((per_plan > 8).mean()), float(per_plan.mean())
for cycle in range(10):
    b = ltp.planSwitchTimesBatch(qg, s0, s1, s2)
    f, m = frac_wide(b, CAP)
    ln = b.traj_len.cpu().numpy()
    print(f"cycle {cycle}: plans with a joint of > 8 runs inside {CAP} samples: {f:.4f}; mean max runs inside {m:.2f}; traj_len <= cap: {(ln[ln>0] <= CAP).mean():.4f} of the live plans, live {(ln>0).mean():.4f}, mean len {ln.mean():.0f}", flush=True)
    s0, s1, s2 = ltp.stateAt(b, 0, N, 100)
