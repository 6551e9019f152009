"""Bring-up script (not collected by pytest): device solvers vs the oracle in det-math mode."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as o
from visual_odometry_amd import ops, synth
dev = torch.device("cuda:0")
o.set_det_math(True)
ok = True
sc = synth.two_view_scene(2000, seed=21)
p1, p2 = sc["p1"], sc["p2"]; pts4 = np.concatenate([p1, p2], 1)
subs = o.gen_subsets(len(p1), 7, 512, p1, p2)
subs2 = ops.gen_subsets(len(p1), 7, 512, pts4)
print("subsets equal", np.array_equal(subs, subs2)); ok &= np.array_equal(subs, subs2)
models, nm = ops.solve_7pt(torch.from_numpy(subs).to(dev), torch.from_numpy(pts4).to(dev))
torch.cuda.synchronize(); models = models.cpu().numpy(); nm = nm.cpu().numpy()
nbit = 0; maxrel = 0; nmis = 0
for h, s in enumerate(subs):
    Fs = o.run7point(p1[s], p2[s])
    if len(Fs) != nm[h]: nmis += 1; continue
    for k in range(nm[h]):
        a = models[h, k].reshape(3, 3); b = Fs[k]
        nbit += np.array_equal(a, b)
        maxrel = max(maxrel, np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
print(f"7pt: count mismatches {nmis}, bit-equal models {nbit}/{nm.sum()}, max rel {maxrel:.3e}", flush=True)
ok &= nmis == 0 and maxrel < 1e-9

ps = synth.pnp_scene(2000, seed=22); obj, img, K = ps["obj"], ps["img"], ps["K"]
subs = o.gen_subsets(len(obj), 4, 512)
pm, rv, nm = ops.solve_p3p(torch.from_numpy(subs).to(dev), torch.from_numpy(obj).to(dev), torch.from_numpy(img).to(dev), K)
torch.cuda.synchronize(); pm = pm.cpu().numpy(); rv = rv.cpu().numpy(); nm = nm.cpu().numpy()
nbit = 0; maxabs = 0; nmis = 0; tot = 0
for h, s in enumerate(subs):
    ns, r, t, R = o.p3p_kernel(obj[s], img[s], K)
    if (ns > 0) != (nm[h] > 0): nmis += 1; continue
    if ns == 0: continue
    tot += 1
    ref = np.concatenate([R.reshape(9), t])
    nbit += np.array_equal(pm[h], ref) and np.array_equal(rv[h], r)
    maxabs = max(maxabs, np.abs(pm[h] - ref).max())
print(f"p3p: count mismatches {nmis}, bit-equal {nbit}/{tot}, max abs {maxabs:.3e}", flush=True)
ok &= nmis == 0 and maxabs < 1e-9

Kinv = np.linalg.inv(sc["K"])
x1 = (np.c_[p1.astype(np.float64), np.ones(len(p1))] @ Kinv.T)[:, :2]
x2 = (np.c_[p2.astype(np.float64), np.ones(len(p2))] @ Kinv.T)[:, :2]
x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
subs = o.gen_subsets(len(p1), 5, 256)
t0 = time.time()
em, nm = ops.solve_5pt(torch.from_numpy(subs).to(dev), torch.from_numpy(x4).to(dev))
torch.cuda.synchronize(); print("5pt gpu time", time.time() - t0)
em = em.cpu().numpy(); nm = nm.cpu().numpy()
nbit = 0; maxabs = 0; nmis = 0; tot = 0
for h, s in enumerate(subs):
    Es = o.run5point(x1[s], x2[s])
    if len(Es) != nm[h]: nmis += 1; continue
    for k in range(nm[h]):
        tot += 1
        a = em[h, k].reshape(3, 3)
        nbit += np.array_equal(a, Es[k]); maxabs = max(maxabs, np.abs(a - Es[k]).max())
print(f"5pt: count mismatches {nmis}, bit-equal {nbit}/{tot}, max abs {maxabs:.3e}", flush=True)
ok &= nmis == 0 and maxabs < 1e-9

# timings
for H in (4096, 65536):
    subs = torch.from_numpy(np.tile(o.gen_subsets(2000, 7, 512, p1, p2), (H // 512, 1))).to(dev)
    P = torch.from_numpy(pts4).to(dev)
    ops.solve_7pt(subs, P); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): ops.solve_7pt(subs, P)
    torch.cuda.synchronize(); print(f"7pt H={H}: {(time.time()-t0)/5*1e6:.0f} us")
    subs = torch.from_numpy(np.tile(o.gen_subsets(2000, 4, 512), (H // 512, 1))).to(dev)
    O = torch.from_numpy(obj).to(dev); I = torch.from_numpy(img).to(dev)
    ops.solve_p3p(subs, O, I, K); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): ops.solve_p3p(subs, O, I, K)
    torch.cuda.synchronize(); print(f"p3p H={H}: {(time.time()-t0)/5*1e6:.0f} us", flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
