"""Diagnostic: k_sample throughput over time inside one launch (block start/end stamps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longtermplanner_amd import LongTermPlanner, limit_set
dof, lim = limit_set("panda")
ltp = LongTermPlanner(dof, 0.001, device=0, **lim)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
qg, q0, v0, a0 = ltp.generateQueries(n)
b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
torch.cuda.synchronize()
off = b.offsets.cpu().numpy().view(np.uint64)
tile = torch.empty(int(off[-1]), dtype=torch.float64, device="cuda")
stamps = torch.zeros(3 * n, dtype=torch.int64, device="cuda")
variant = sys.argv[2] if len(sys.argv) > 2 else "0"
L = b.traj_len.cpu().numpy().astype(np.int64)
byt = 32 * dof * L
ltp.sampleBatch(b, 0, n, tile, spread=int(variant))
torch.cuda.synchronize()
ltp._lib.ltp_debug_set_sample_stamps(ltp._h, stamps.data_ptr())
for rep in range(3):
    ltp.sampleBatch(b, 0, n, tile, spread=int(variant))
    torch.cuda.synchronize()
    st3 = stamps.cpu().numpy().reshape(n, 3)
    st = st3[:, [0, 2]]
    live = st3[:, 2] > 0
    print(f"   table build: mean {((st3[live, 1] - st3[live, 0]) / 100).mean():.1f} us of a mean block lifetime {((st3[live, 2] - st3[live, 0]) / 100).mean():.1f} us")
    t0 = st[:, 0].min(); t1 = st[:, 1].max()
    bins = ((st[:, 1] - t0) // 100000).astype(int)
    agg = np.bincount(bins, weights=byt)
    print(f"rep {rep}: {(t1 - t0) / 1e5:.2f} ms, {byt.sum() / ((t1 - t0) / 1e8) / 1e9:.0f} GB/s; per-ms bins:", (np.round(agg / 1e-3 / 1e11) ).astype(int).tolist())
    # same, binned by ADDRESS (GB offset of the block in the tile) instead of time
    gb = (off[:-1].astype(np.float64) * 8 / 1e9 / 8).astype(int)      # 8 GB address bins
    dur = (st[:, 1] - st[:, 0]).astype(np.float64)
    life = np.bincount(gb, weights=dur) / np.bincount(gb)
    print("   mean block lifetime (us) per 8 GB of tile address:", np.round(life / 100).astype(int).tolist())
