"""Bring-up script (not collected by pytest): first GPU contact for K1-K3 against the oracle."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as o
from visual_odometry_amd import ops, synth

dev = torch.device("cuda:0")
print("device", torch.cuda.get_device_name(0), flush=True)

def check_knn(nq, nt, seed):
    q, t = synth.sift_like_pair(nq, nt, seed)
    if nq >= 64 and nt >= 64:   # forced exact ties and duplicates
        t[5] = t[3]; t[40] = t[3]; q[7] = t[3]
        q[9] = 0; t[11] = 255
    qi8, qn, f1 = ops.desc_to_i8(torch.from_numpy(q).to(dev))
    ti8, tn, f2 = ops.desc_to_i8(torch.from_numpy(t).to(dev))
    idx, d2 = ops.knn2(qi8, qn, ti8, tn)
    torch.cuda.synchronize()
    idx = idx.cpu().numpy(); d2 = d2.cpu().numpy()
    t0 = time.time(); oi, od = o.knn2(q, t, mt=True); t1 = time.time()
    D = (od.astype(np.float64) ** 2)
    ok_i = np.array_equal(idx, oi)
    ok_d = np.array_equal(np.sqrt(d2.astype(np.float32)).astype(np.float32), od)
    print(f"knn {nq}x{nt}: idx equal {ok_i}, dist equal {ok_d}, flags {f1.item()} {f2.item()}, oracle {t1-t0:.2f}s", flush=True)
    if not ok_i:
        bad = np.nonzero((idx != oi).any(1))[0]
        print("  mismatches", len(bad), bad[:5], idx[bad[:5]], oi[bad[:5]], d2[bad[:5]], od[bad[:5]]**2)
    return ok_i and ok_d

ok = True
for nq, nt, seed in [(64, 64, 1), (100, 37, 2), (1, 2, 3), (300, 1000, 4), (1024, 1024, 5), (4096, 4096, 6), (8192, 8192, 7)]:
    ok &= check_knn(nq, nt, seed)

# scoring
sc = synth.two_view_scene(3000, seed=11)
rng = np.random.default_rng(0)
p1, p2 = sc["p1"], sc["p2"]
subs = o.gen_subsets(len(p1), 7, 300, p1, p2)
models = []
for s in subs:
    Fs = o.run7point(p1[s], p2[s])
    for F in Fs: models.append(F.reshape(9))
models = np.array(models)
pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
for thr in (0.1, 1.0):
    cnt = ops.score_fund(torch.from_numpy(models).to(dev), torch.from_numpy(pts4).to(dev), thr).cpu().numpy()
    ref = np.array([o.find_inliers(o.fund_error(m, p1, p2), thr)[0] for m in models])
    print(f"score_fund thr {thr}: H={len(models)} equal {np.array_equal(cnt, ref)} max {cnt.max()} {ref.max()}", flush=True)
    ok &= np.array_equal(cnt, ref)
    b = int(np.argmax(ref))
    m = ops.mask_fund(torch.from_numpy(models[b]).to(dev), torch.from_numpy(pts4).to(dev), thr).cpu().numpy()
    ok &= np.array_equal(m, o.find_inliers(o.fund_error(models[b], p1, p2), thr)[1])

ps = synth.pnp_scene(3000, seed=12)
obj, img, K = ps["obj"], ps["img"], ps["K"]
subs = o.gen_subsets(len(obj), 4, 500)
pm = []
for s in subs:
    ns, rv, tv, R = o.p3p_kernel(obj[s], img[s], K)
    if ns > 0: pm.append(np.concatenate([R.reshape(9), tv]))
pm = np.array(pm)
for thr in (1.0, 5.0):
    cnt = ops.score_pnp(torch.from_numpy(pm).to(dev), torch.from_numpy(obj).to(dev), torch.from_numpy(img).to(dev), K, thr).cpu().numpy()
    ref = np.array([o.find_inliers(o.pnp_error(m[:9], m[9:], K, obj, img), thr)[0] for m in pm])
    print(f"score_pnp thr {thr}: H={len(pm)} equal {np.array_equal(cnt, ref)} max {cnt.max()} {ref.max()}", flush=True)
    ok &= np.array_equal(cnt, ref)

# essential
Kinv = np.linalg.inv(sc["K"])
x1 = (np.c_[p1.astype(np.float64), np.ones(len(p1))] @ Kinv.T)[:, :2]
x2 = (np.c_[p2.astype(np.float64), np.ones(len(p2))] @ Kinv.T)[:, :2]
subs = o.gen_subsets(len(p1), 5, 100)
em = []
for s in subs:
    for E in o.run5point(x1[s], x2[s]): em.append(E.reshape(9))
em = np.array(em)
x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
thr = 1.0 / 718.856
cnt = ops.score_ess(torch.from_numpy(em).to(dev), torch.from_numpy(x4).to(dev), thr).cpu().numpy()
ref = np.array([o.find_inliers(o.sampson_error(m, x1, x2), thr)[0] for m in em])
print(f"score_ess: H={len(em)} equal {np.array_equal(cnt, ref)} max {cnt.max()} {ref.max()}", flush=True)
ok &= np.array_equal(cnt, ref)

# replay
cnts = rng.integers(0, 50, size=(1000, 3)).astype(np.int32); cnts[200, 1] = 900; cnts[700, 0] = 2500
nm = rng.integers(0, 4, size=1000).astype(np.int32); nm[200] = 3; nm[700] = 3
res = ops.ransac_replay(torch.from_numpy(cnts).to(dev), torch.from_numpy(nm).to(dev), 3000, 7, 0.9999, 1000).cpu().numpy()
print("replay", res, o.ransac_replay(cnts, nm, 3000, 7, 0.9999, 1000), flush=True)
ok &= tuple(res) == o.ransac_replay(cnts, nm, 3000, 7, 0.9999, 1000)

# timing knn 8192
q, t = synth.sift_like_pair(8192, 8192, 7)
qi8, qn, _ = ops.desc_to_i8(torch.from_numpy(q).to(dev)); ti8, tn, _ = ops.desc_to_i8(torch.from_numpy(t).to(dev))
ws = torch.empty((64 << 20,), dtype=torch.uint8, device=dev)
for _ in range(5): ops.knn2(qi8, qn, ti8, tn, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.knn2(qi8, qn, ti8, tn, ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"knn 8192^2: {ms*1000:.1f} us/call -> {2*8192*8192*128/ms/1e9:.1f} TOPS", flush=True)
mt = torch.from_numpy(np.tile(models, (int(np.ceil(4096/len(models))), 1))[:4096].copy()).to(dev)
p8 = torch.from_numpy(np.tile(pts4, (3, 1))[:8192].copy()).to(dev)
for _ in range(3): ops.score_fund(mt, p8, 0.1)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ops.score_fund(mt, p8, 0.1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"score_fund 4096x8192: {ms*1000:.1f} us -> {4096*8192/ms/1e6:.2f} G evals/s, eff {4096*8192*16/ms/1e9:.2f} TB/s", flush=True)
print("ALL OK" if ok else "FAILURES", flush=True)
sys.exit(0 if ok else 1)
