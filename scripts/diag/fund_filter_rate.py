"""Diagnostic (GPU box): how often does the F-scoring filter say "cannot tell" on the bench workload?
Emulates fund_filter in numpy (no FMA; the margins dwarf that) on the models the 7-point kernel makes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, synth, pipeline
import bench

dev = torch.device("cuda:0")
ps = bench.PairSet(1, dev, 0, 1)
r = bench.step(ps)
torch.cuda.synchronize()
n = int(r.n_match[0])
pts4 = r.pts4[0, :n].cpu().numpy().astype(np.float64)
samples = ops.gen_subsets_device(r.n_match[:1], r.pts4.shape[1], 7, 512, r.pts4[:1])[0][0]
models, nm = ops.solve_7pt(samples, r.pts4[0])
models = models.cpu().numpy(); nm = nm.cpu().numpy()
F = np.array([models[h, k] for h in range(len(nm)) for k in range(nm[h])])
print("models", F.shape, "points", n)
t = float(np.float32(bench.F_THR ** 2))
x1, y1, x2, y2 = pts4.T
Q1 = max(1.0, np.abs(pts4[:, :2]).max()); Q2 = max(1.0, np.abs(pts4[:, 2:]).max())
A = np.abs(F)
S = (A[:, 0] + A[:, 1] + A[:, 3] + A[:, 4]) * Q1 * Q2 + (A[:, 2] + A[:, 5]) * Q2 + (A[:, 6] + A[:, 7]) * Q1 + A[:, 8]
eta = 2.0 ** -44 * S
al2 = 2.0 ** -46 * np.maximum((A[:, 0] + A[:, 1]) * Q1 + A[:, 2], (A[:, 3] + A[:, 4]) * Q1 + A[:, 5])
al1 = 2.0 ** -46 * np.maximum((A[:, 0] + A[:, 3]) * Q2 + A[:, 6], (A[:, 1] + A[:, 4]) * Q2 + A[:, 7])
c1 = 2 * (eta ** 2 + 2.0 ** 24 * t * al1 ** 2) + 2.0 ** -600
c2 = 2 * (eta ** 2 + 2.0 ** 24 * t * al2 ** 2) + 2.0 ** -600
f = F[:, :, None]
a1 = f[:, 0] * x1 + f[:, 1] * y1 + f[:, 2]; b1 = f[:, 3] * x1 + f[:, 4] * y1 + f[:, 5]; cc1 = f[:, 6] * x1 + f[:, 7] * y1 + f[:, 8]
den2 = a1 * a1 + b1 * b1; d2 = x2 * a1 + y2 * b1 + cc1
a2 = f[:, 0] * x2 + f[:, 3] * y2 + f[:, 6]; b2 = f[:, 1] * x2 + f[:, 4] * y2 + f[:, 7]; cc2 = f[:, 2] * x2 + f[:, 5] * y2 + f[:, 8]
den1 = a2 * a2 + b2 * b2; d1 = x1 * a2 + y1 * b2 + cc2
z1 = d1 * d1 - t * den1; z2 = d2 * d2 - t * den2
m1 = 2 * eta[:, None] * np.abs(d1) + 2.0 ** -19 * t * den1 + c1[:, None]
m2 = 2 * eta[:, None] * np.abs(d2) + 2.0 ** -19 * t * den2 + c2[:, None]
inl = (z1 < -m1) & (z2 < -m2); out = (z1 > m1) | (z2 > m2)
amb = ~(inl | out)
print("amb fraction overall", amb.mean(), "models with any amb", amb.any(1).mean(), "points with any amb", amb.any(0).mean())
pm = amb.mean(1)
worst = np.argsort(pm)[::-1][:8]
for w in worst:
    print("model", w, "amb frac", pm[w], "F", F[w], "S", S[w], "eta", eta[w], "median t*den1", np.median(t * den1[w]), "c1", c1[w], "median |d1|", np.median(np.abs(d1[w])))
# term shares for ambiguous entries
i, j = np.nonzero(amb)
if len(i):
    k = slice(0, 10)
    print("sample amb entries: z1", z1[i[k], j[k]], "m1", m1[i[k], j[k]], "2eta|d|", (2 * eta[i] * np.abs(d1[i, j]))[k], "rel", (2.0 ** -19 * t * den1[i, j])[k], "c1", c1[i[k]])
wave = amb[: len(F) // 64 * 64].reshape(-1, 64, amb.shape[1]).any(1)
print("per wave-eval amb rate", wave.mean())

# device-side count of fallback groups (needs VO_HIP_LIB=ablate/libvo_hip_sabl2.so)
if "sabl2" in os.environ.get("VO_HIP_LIB", ""):
    cnt = ops.score_fund(torch.from_numpy(F).to(dev), r.pts4[0, :n].contiguous(), bench.F_THR).cpu().numpy()
    own = cnt >> 22; wave = (cnt >> 12) & 1023
    groups = n // 8
    print("device: groups per model", groups, "own-amb groups mean", own.mean(), "wave-amb groups mean", wave.mean(),
          "max", wave.max(), "models with own amb", (own > 0).sum(), "of", len(F))
    bad = np.nonzero(own > 0)[0][:5]
    for b_ in bad:
        print("bad model", b_, own[b_], F[b_])
# early-exit statistics: a wave of 64 consecutive models skips the second denominator for a correspondence
# when every lane is already "sure outlier" against den2 alone
notout = ~(d2 * d2 > t * den2 * (1 + 2.0 ** -16))
print("not-out-by-den2 fraction", notout.mean(), "inlier fraction", inl.mean())
G = (len(F) // 64) * 64
w = notout[:G].reshape(-1, 64, notout.shape[1]).any(1)
print("wave x correspondence that needs the full test:", w.mean())
per_model = notout.mean(1)
print("per-model not-out fraction: median", np.median(per_model), "p90", np.quantile(per_model, 0.9), "p99", np.quantile(per_model, 0.99), "max", per_model.max())
# the same with the correspondences across the lanes (score_fund_kernel): 64 consecutive correspondences, one model
Gp = (notout.shape[1] // 64) * 64
wp = notout[:, :Gp].reshape(len(F), -1, 64).any(2)
print("model x 64-correspondence group that needs the full test:", wp.mean())
f32amb = ~(inl | out)      # f64-filter ambiguity, a lower bound for the f32 tier
