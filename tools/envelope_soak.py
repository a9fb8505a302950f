"""Envelope consumer against the reduction of the dense rows, on the GPU, bit for bit: for every plan the min / max of q over windows of
`window` samples of the rows ltp_sample_batch stores must equal ltp_envelope_batch's output exactly.
usage (on the GPU box): python tools/envelope_soak.py [plans_per_set] [window] [n_windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import longtermplanner_amd as amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
window = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 40
total_w = total_bad = 0
worst = 0.0
for name, seed in (("panda", 101), ("ref", 102), ("ref30", 103), ("panda", 104)):
    D, lim = amd.limit_set(name)
    nn = n if D <= 7 else n // 5
    ltp = amd.LongTermPlanner(D, 0.001, device=0, **lim)
    q = ltp.generateQueries(nn, seed=seed)
    if seed == 104:                                   # rest-to-rest moves: v_0 = a_0 = 0 (the derivative of q vanishes at the first sample)
        q[2].zero_(); q[3].zero_()
    b = ltp.planSwitchTimesBatch(*q)
    env = ltp.envelopeBatch(b, 0, nn, window, nwin)
    off = b.offsets.cpu().numpy().view(np.uint64)
    lens = b.traj_len.cpu().numpy()
    chunk = 4000
    for f in range(0, nn, chunk):
        c = min(chunk, nn - f)
        tile = torch.zeros(int(off[f + c] - off[f]), dtype=torch.float64, device="cuda")
        ltp.sampleBatch(b, f, c, tile)
        torch.cuda.synchronize()
        for i in range(c):
            p = f + i
            L = int(lens[p])
            if L == 0:
                continue
            stride = (L + 31) // 32 * 32
            base = int(off[p] - off[f])
            rows = tile[base: base + D * stride].view(D, stride)[:, :L]                 # the q rows
            need = window * nwin
            if L < need:
                rows = torch.cat([rows, rows[:, -1:].expand(D, need - L)], dim=1)
            w = rows[:, :need].reshape(D, nwin, window)
            lo, hi = w.min(dim=2).values, w.max(dim=2).values
            e = env[p]
            bad = int((e[..., 0] != lo).sum().item() + (e[..., 1] != hi).sum().item())
            total_w += 2 * D * nwin
            if bad:
                total_bad += bad
                worst = max(worst, float((e[..., 0] - lo).abs().max().item()), float((e[..., 1] - hi).abs().max().item()))
    print(f"{name} seed {seed}: {nn} plans, {window} x {nwin} windows: {total_w} values compared so far, {total_bad} not bit-identical, worst |d| {worst:.3e}", flush=True)
