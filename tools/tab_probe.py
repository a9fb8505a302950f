"""Diagnostic: where the time of an item goes in the table-pass sampler (k_sample_tab_*): loader and streaming-wave stamps.
usage: python tools/tab_probe.py [n] [max_samples] [f32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longtermplanner_amd import LongTermPlanner, limit_set
dof, lim = limit_set("panda")
ltp = LongTermPlanner(dof, 0.001, device=0, **lim)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 256
f32 = len(sys.argv) > 3
ltp.setMaxSamples(cap)
ltp.setTablePass(1, 16 << 30)
qg, q0, v0, a0 = ltp.generateQueries(n)
b = ltp.planSwitchTimesBatch(qg, q0, v0, a0)
torch.cuda.synchronize()
total = int(b.offsets[-1].item())
tile = torch.empty(total, dtype=torch.float32 if f32 else torch.float64, device="cuda")
ngroups = (dof + 6) // 7
per = (n + 63) // 64
items = per * 64 * ngroups
stamps = torch.zeros(8 * items, dtype=torch.int64, device="cuda")
ltp.sampleBatch(b, 0, n, tile, tables=True)
torch.cuda.synchronize()
ltp._lib.ltp_debug_set_sample_stamps(ltp._h, stamps.data_ptr())
for rep in range(2):
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ltp.sampleBatch(b, 0, n, tile, tables=True); e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(items, 8).astype(np.float64) / 100.0     # us
    ok = (s[:, 1] > 0) & (s[:, 3] > 0) & (s[:, 4] > 0) & (s[:, 5] > 0)
    s = s[ok]
    print(f"rep {rep}: {e0.elapsed_time(e1):.2f} ms (build + sample), {ok.sum()} items stamped; per item, mean us:")
    print(f"   loader: loads in -> item published {np.mean(s[:,3]-s[:,1]):.2f}   streaming wave 0: published -> starts {np.mean(s[:,4]-s[:,3]):.2f}, "
          f"rows issued in {np.mean(s[:,5]-s[:,4]):.2f}")
    okl = s[:, 7] > 0
    print(f"   loader iteration: wait for a free buffer {np.mean((s[:,0]-s[:,7])[okl]):.2f}, draw + request {np.mean((s[:,6]-s[:,0])[okl]):.2f}, "
          f"wait for the loads {np.mean((s[:,1]-s[:,6])[okl]):.2f}, publish {np.mean((s[:,3]-s[:,1])[okl]):.2f}")
    print(f"   items whose tables needed the second fetch (more than 8 runs of a joint inside the cap): {int((s[:, 2] > 0).sum())}")
    t = np.sort(s[:, 3])
    print(f"   span {(s[:,5].max()-s[:,1].min())/1e3:.2f} ms; items published per us over the middle half: {len(t)/2/(t[3*len(t)//4]-t[len(t)//4]):.1f}")
ltp._lib.ltp_debug_set_sample_stamps(ltp._h, None)
