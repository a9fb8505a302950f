"""Analytic envelopes (ltp_set_envelope_mode LTP_ENVELOPE_ANALYTIC) against the exhaustive ones (every sample evaluated: the bits of the
reduced rows), on the GPU: the same planned batch through both modes, every window value compared. VERDICT r5 item 7: >= 1e9 window
values over the panda limits, the reference's, 30-DoF and wide-fuzzed limit sets before the default may change.
usage (on the GPU box): python tools/envelope_mode_soak.py [plans_per_big_set] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import longtermplanner_amd as amd
from dense_compare import fuzz_limits

n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
sets = []
for name, seed, geo in (("panda", 201, (64, 32)), ("ref", 202, (64, 32)), ("panda", 203, (256, 8)), ("ref", 204, (16, 128)), ("ref30", 205, (64, 32)), ("panda", 206, (7, 300))):
    D, lim = amd.limit_set(name)
    sets.append((f"{name}:{geo[0]}x{geo[1]}", D, 0.001, lim, n_big if D <= 7 else n_big // 5, seed, geo, False))
rng = np.random.default_rng(2026)
for trial in range(36):
    D, ts, lim = fuzz_limits(rng, trial, True)
    sets.append((f"fuzz{trial}", D, ts, lim, 60_000, 9100 + trial, (int(rng.choice([8, 64, 200])), int(rng.choice([16, 40, 100]))), trial % 4 == 0))
total = differing = 0
worst = 0.0
worst_rel = 0.0
lines = []
for name, D, ts, lim, n, seed, (window, nwin), rest in sets:
    ltp = amd.LongTermPlanner(D, ts, device=0, **lim)
    q = [x.clone() for x in ltp.generateQueries(n, seed=seed)]
    if rest:                                          # rest-to-rest moves: q'(m) vanishes at the first sample
        q[2].zero_(); q[3].zero_()
    b = ltp.planSwitchTimesBatch(*q)
    ltp.setEnvelopeMode("exhaustive")
    ex = ltp.envelopeBatch(b, 0, n, window, nwin).clone()
    st_ex = b.status.clone()
    b2 = ltp.planSwitchTimesBatch(*q)
    ltp.setEnvelopeMode("analytic")
    an = ltp.envelopeBatch(b2, 0, n, window, nwin)
    torch.cuda.synchronize()
    assert torch.equal(b2.status, st_ex), name
    both_nan = torch.isnan(ex) & torch.isnan(an)
    d = torch.where(both_nan, torch.zeros_like(ex), (ex - an).abs())
    assert not bool(torch.isnan(d).any()), name        # NaN in one mode only
    nd = int((d != 0).sum().item())
    w = float(d.max().item())
    scale = float(torch.nan_to_num(ex).abs().max().item())
    total += ex.numel(); differing += nd; worst = max(worst, w)
    line = {"set": name, "dof": D, "t_sample": ts, "plans": n, "window": window, "n_windows": nwin, "values": ex.numel(), "not_bit_identical": nd, "max_abs_d": w, "max_abs_q": scale}
    lines.append(line)
    print(json.dumps(line), flush=True)
summary = {"values_compared": total, "not_bit_identical": differing, "max_abs_d": worst, "bar": 1e-12, "within_bar": worst <= 1e-12, "sets": len(sets)}
print(json.dumps(summary))
if out_path:
    json.dump({"summary": summary, "sets": lines, "command": "python tools/envelope_mode_soak.py " + " ".join(sys.argv[1:])}, open(out_path, "w"), indent=1)
