"""GPU parity: RANSAC scoring (K2/K3), masks (K5), replay/argmax (K6), minimal solvers (K4) and the
end-to-end drop-in entry points, against the oracle.  Counts, masks, replay decisions and (in the
oracle's det-math mode) solver outputs are compared bit for bit; final poses to 1e-6 relative."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import cv2_shim, ops, params, ransac, synth
from visual_odometry_amd.pnp import pnp
from visual_odometry_amd.utils import geometry

pytestmark = pytest.mark.gpu


def _d(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def fund_models(oracle_mod):
    sc = synth.two_view_scene(3000, seed=11)
    p1, p2 = sc["p1"], sc["p2"]
    subs = oracle_mod.gen_subsets(len(p1), 7, 300, p1, p2)
    models = np.array([F.reshape(9) for s in subs for F in oracle_mod.run7point(p1[s], p2[s])])
    return sc, p1, p2, models


@pytest.mark.parametrize("thr", [0.1, 1.0, 3.0])
def test_score_and_mask_fund(oracle_mod, dev, fund_models, thr):
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    cnt = ops.score_fund(_d(models, dev), _d(pts4, dev), thr).cpu().numpy()
    ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), thr)[0] for m in models])
    assert np.array_equal(cnt, ref)
    for b in (int(np.argmax(ref)), 0, len(models) - 1):
        m = ops.mask_fund(_d(models[b], dev), _d(pts4, dev), thr).cpu().numpy()
        assert np.array_equal(m, oracle_mod.find_inliers(oracle_mod.fund_error(models[b], p1, p2), thr)[1])


def test_score_fund_threshold_edge(oracle_mod, dev, fund_models):
    """thresholds placed exactly on, one ulp below and one ulp above an error value."""
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    m = models[int(len(models) / 2)]
    err = oracle_mod.fund_error(m, p1, p2)
    e = np.sort(err)[len(err) // 3]
    for t2 in (e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(np.inf))):
        thr = float(np.sqrt(np.float64(t2)))
        # only use thresholds whose (float)(thr*thr) lands back on the intended f32 value
        if np.float32(thr * thr) != t2:
            continue
        cnt = ops.score_fund(_d(m[None], dev), _d(pts4, dev), thr).cpu().numpy()
        assert cnt[0] == int((err <= t2).sum())


def test_score_fund_filter_at_every_boundary(oracle_mod, dev, fund_models):
    """The count kernel decides most correspondences with a division-free filter and falls back to the
    exact evaluation near the threshold: put the threshold ON the error of many different points (and
    one f32 ulp either side) for the best and for a mediocre model."""
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    counts = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), 1.0)[0] for m in models])
    checked = 0
    for m in (models[int(np.argmax(counts))], models[len(models) // 2]):
        err = oracle_mod.fund_error(m, p1, p2)
        for e in np.sort(err)[5:1500:37]:
            for t2 in (e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(np.inf))):
                thr = float(np.sqrt(np.float64(t2)))
                if np.float32(thr * thr) != t2:
                    continue
                cnt = ops.score_fund(_d(m[None], dev), _d(pts4, dev), thr).cpu().numpy()
                assert cnt[0] == int((err <= t2).sum())
                checked += 1
    assert checked > 40


def test_score_fund_filter_degenerate_and_scaled(oracle_mod, dev, fund_models):
    """Zero / NaN / Inf / denormal-scale models and far-away or non-finite points: the filter must hand
    every such case to the exact evaluation (same counts as the scalar definition)."""
    sc, p1, p2, models = fund_models
    rng = np.random.default_rng(5)
    base = models[:24].copy()
    extra = [np.zeros(9), np.full(9, np.nan), np.full(9, np.inf)]
    for k in range(9):
        m = base[k].copy(); m[k] = np.nan; extra.append(m)
        m = base[k].copy(); m[k] = np.inf; extra.append(m)
        m = base[k].copy(); m[k] = 0.0; extra.append(m)
    m = np.zeros(9); m[8] = 1.0; extra.append(m)                      # a = b = 0: zero denominators
    m = np.zeros(9); m[2] = 1.0; m[6] = -1.0; extra.append(m)
    scaled = [base[i] * s for i, s in enumerate([1e-300, 1e-160, 1e-150, 1e-100, 1e100, 1e150, 1e160, 1e300])]
    mods = np.concatenate([base, np.array(extra), np.array(scaled)])
    for scale, poison in ((1.0, False), (1.0, True), (1e4, False), (1e-3, False)):
        q1 = (p1[:700].astype(np.float64) * scale).astype(np.float32)
        q2 = (p2[:700].astype(np.float64) * scale).astype(np.float32)
        if poison:
            q1[3, 0] = np.nan; q2[10, 1] = np.inf; q1[20] = 0; q2[20] = 0; q1[21, 0] = 3e38
        pts4 = np.concatenate([q1, q2], 1).astype(np.float32)
        for thr in (0.1 * scale, 2.0 * scale):
            with np.errstate(all="ignore"):
                ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, q1, q2), thr)[0] for m in mods])
            cnt = ops.score_fund(_d(mods, dev), _d(pts4, dev), thr).cpu().numpy()
            assert np.array_equal(cnt, ref), (scale, poison, thr, np.nonzero(cnt != ref)[0][:10])


@pytest.mark.parametrize("seed", range(6))
def test_score_fund_fuzz(oracle_mod, dev, seed):
    """random scenes, image scales, model scalings and thresholds (from 0.01 px to 30 px): counts equal
    the scalar definition whatever tier of the filter decides."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(33, 2500))
    sc = synth.two_view_scene(n, seed=200 + seed, outlier_frac=float(rng.uniform(0, 0.6)), noise_px=float(rng.uniform(0, 2)))
    scale = float(10 ** rng.uniform(-2, 2))
    p1 = (sc["p1"] * scale).astype(np.float32); p2 = (sc["p2"] * scale).astype(np.float32)
    subs = oracle_mod.gen_subsets(n, 7, 150, p1, p2)
    models = np.array([F.reshape(9) for s_ in subs for F in oracle_mod.run7point(p1[s_], p2[s_])])
    models = models * (10 ** rng.uniform(-30, 30, (len(models), 1)))          # F is defined up to scale
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    for thr in 10 ** rng.uniform(-2, 1.5, 4) * scale:
        cnt = ops.score_fund(_d(models, dev), _d(pts4, dev), float(thr)).cpu().numpy()
        ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), float(thr))[0] for m in models])
        assert np.array_equal(cnt, ref), (seed, thr, np.nonzero(cnt != ref)[0][:5])


@pytest.mark.parametrize("seed", range(8))
def test_score_fund_mixed_dynamic_range_and_unbalanced_models(oracle_mod, dev, seed):
    """What the scene fuzz does not produce (scripts/diag/fscore_fuzz_adversarial.py ran 200 seeds of it): a few far points set
    the problem-wide power-of-two scaling (bound up to 2^19) and everything else is tiny against it, so the f16 lo halves of
    the matrix-core tier's split coordinates go subnormal or vanish; points at the origin; models with entries of unbalanced
    magnitude, sparse and rank-one models; thresholds from 1e-3 to 1e2 of the image scale.  Counts equal the oracle's."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(40, 1500))
    W = float(10 ** rng.uniform(-1, 3))
    sc = synth.two_view_scene(n, seed=900 + seed, outlier_frac=float(rng.uniform(0, 0.5)), noise_px=float(rng.uniform(0, 1)))
    p1 = sc["p1"] * (W / 1024.0); p2 = sc["p2"] * (W / 1024.0)
    subs = oracle_mod.gen_subsets(n, 7, 30, p1.astype(np.float32), p2.astype(np.float32))
    models = [F.reshape(9) for s_ in subs for F in oracle_mod.run7point(p1[s_].astype(np.float32), p2[s_].astype(np.float32))]
    if seed % 4:
        far = rng.choice(n, size=max(1, n // 50), replace=False)
        big = 2.0 ** rng.uniform(4, 19 - np.log2(max(W, 1.0)), size=(len(far), 1))
        if seed % 4 in (1, 3): p1[far] = p1[far] * big
        if seed % 4 in (2, 3): p2[far] = p2[far] * big
    if seed % 4 == 2:
        z = rng.choice(n, size=5, replace=False)
        p1[z[:3]] = 0.0; p2[z[2:]] = rng.uniform(-1e-6, 1e-6, (3, 2)) * W
    p1 = p1.astype(np.float32); p2 = p2.astype(np.float32)
    for _ in range(30):
        models.append((rng.standard_normal((3, 3)) * 10 ** rng.uniform(-3, 3, (3, 3))).reshape(9))
    for _ in range(8):
        models.append((rng.standard_normal((3, 3)) * (rng.uniform(0, 1, (3, 3)) < 0.5)).reshape(9))
    for _ in range(8):
        models.append(np.outer(rng.standard_normal(3), rng.standard_normal(3)).reshape(9))
    models = np.array(models, dtype=np.float64)
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    for thr in 10 ** rng.uniform(-3, 2, 3) * W:
        cnt = ops.score_fund(_d(models, dev), _d(pts4, dev), float(thr)).cpu().numpy()
        ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), float(thr))[0] for m in models])
        assert np.array_equal(cnt, ref), (seed, thr, W, np.nonzero(cnt != ref)[0][:5])


@pytest.mark.parametrize("thr", [0.1, 1.0, 5.0])
def test_score_fund_bound_is_never_below_the_count(oracle_mod, dev, fund_models, thr):
    """vo_score_fund_bound (the matrix-core tier alone, what vo_ransac_batch prunes hopeless hypotheses with): an upper bound
    of every model's inlier count -- also for zero / NaN / absurdly scaled models and for a problem with non-finite
    coordinates, where nothing can be proven and the bound is the number of correspondences -- and a tight one on scene data."""
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    odd = np.array([np.zeros(9), np.full(9, np.nan), models[0] * 1e300, models[1] * 1e-300, np.full(9, np.inf)])
    allm = np.concatenate([models, odd])
    for n in (len(p1), 777, 65, 1):
        ub = ops.score_fund_bound(_d(allm, dev), _d(pts4[:n], dev), thr).cpu().numpy()
        cnt = ops.score_fund(_d(allm, dev), _d(pts4[:n], dev), thr).cpu().numpy()
        assert np.all(ub >= cnt) and np.all(ub <= n), (thr, n, np.nonzero(ub < cnt)[0][:5])
        assert np.all(ub[len(models):len(models) + 2] == n) and ub[-1] == n     # zero / NaN / Inf model: nothing proven
        if n == len(p1):
            assert (ub[:len(models)] - cnt[:len(models)]).sum() <= 0.5 * cnt[:len(models)].sum() + 0.01 * n * len(models)   # (den2 alone + the f16 band: ~3 % above a good model's count)
    bad = pts4.copy(); bad[5, 2] = np.inf
    ub = ops.score_fund_bound(_d(models, dev), _d(bad, dev), thr).cpu().numpy()
    assert np.all(ub == len(bad))


@pytest.mark.parametrize("thr,first_block", [(0.1, 8), (0.1, 100), (1.0, 37), (3.0, 64), (30.0, 16)])
def test_ransac_batch_fund_pruned_blocks_equal_one_block(dev, thr, first_block):
    """Blocks behind the first count only the hypotheses whose upper bound beats the best count so far (csrc/vo_score.hip:
    vo_launch_score_fund_pruned): result record, F and mask are those of the single exact block, bit for bit -- problems
    of different sizes in one batch (n_dev), one with every correspondence repeated four times (ties), one with fewer than
    15 correspondences, one with non-finite coordinates, a threshold so wide that every model is a contender."""
    B, H, N = 6, 1500, 2400
    rng = np.random.default_rng(int(thr * 10) + first_block)
    pts = np.zeros((B, N, 4), np.float32); nd = np.zeros(B, np.int32); subs = np.zeros((B, H, 7), np.int32)
    for b in range(B):
        n = [2400, 300, 1000, 15, 12, 2000][b]
        sc = synth.two_view_scene(n, seed=500 + b, outlier_frac=float(rng.uniform(0.1, 0.6)), noise_px=float(rng.uniform(0.05, 1.0)))
        p = np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)
        if b == 2:
            p = np.tile(p[:250], (4, 1))
        pts[b, :n] = p; nd[b] = n
        if n >= 15:
            subs[b] = ops.gen_subsets(n, 7, H, p)
    nd[4] = 0                                                   # (a problem without correspondences: no model)
    pts[5, 7, 1] = np.inf; pts[5, 11, 2] = np.nan              # non-finite coordinates: nothing can be bounded, everybody is a contender
    outs = []
    for fb in (0, first_block):
        res, best, _, mask = ops.ransac_batch(ops.KIND_FUND, _d(subs, dev), _d(pts, dev), thr, 0.9999, H, n_dev=_d(nd, dev),
                                              first_block=fb)
        outs.append((res.cpu().numpy(), best.cpu().numpy(), mask.cpu().numpy()))
    (r0, f0, m0), (r1, f1, m1) = outs
    assert np.array_equal(r0, r1), (r0, r1)
    for b in range(B):
        if r0[b, 0] >= 0:
            assert np.array_equal(f0[b], f1[b]) and np.array_equal(m0[b, :nd[b]], m1[b, :nd[b]]), b
    assert r0[0, 2] > 7 and r0[0, 0] >= 0                       # (the runs did find models)


@pytest.mark.parametrize("n", [1000, 333, 64 * 5 + 1])
def test_score_fund_many_models_walk_chunks(oracle_mod, dev, fund_models, n):
    """The F count kernel's workgroups walk several chunks of 256 correspondences once there are enough model blocks
    (csrc/vo_score_fmfma.h): 140 000 model rows (the fixture's models repeated) over point counts that leave a ragged
    last chunk and a ragged last group of 64; every row's count equals the oracle's for its model."""
    sc, p1, p2, models = fund_models
    p1, p2 = p1[:n], p2[:n]
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    rows = 140_000
    rep = models[np.arange(rows) % len(models)]
    cnt = ops.score_fund(_d(rep, dev), _d(pts4, dev), 0.5).cpu().numpy()
    ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), 0.5)[0] for m in models])
    assert np.array_equal(cnt, ref[np.arange(rows) % len(models)])


@pytest.mark.parametrize("n", [1, 7, 40, 64, 65, 255, 256, 257])
def test_score_fund_few_correspondences(oracle_mod, dev, fund_models, n):
    """Point counts around the group (64) and chunk (256) sizes of the MFMA tier, down to a single correspondence."""
    sc, p1, p2, models = fund_models
    p1, p2 = p1[:n], p2[:n]
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    cnt = ops.score_fund(_d(models, dev), _d(pts4, dev), 1.0).cpu().numpy()
    ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), 1.0)[0] for m in models])
    assert np.array_equal(cnt, ref)


def test_score_fund_every_evaluation_survives_the_first_tier(oracle_mod, dev, fund_models):
    """A threshold as large as the image: (nearly) every evaluation passes the MFMA tier's sure-outlier test, so the
    per-wave survivor queue overflows in every group and the lanes work their own survivor strings off."""
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    for thr in (400.0, 3000.0):
        cnt = ops.score_fund(_d(models[:200], dev), _d(pts4, dev), thr).cpu().numpy()
        ref = np.array([oracle_mod.find_inliers(oracle_mod.fund_error(m, p1, p2), thr)[0] for m in models[:200]])
        assert np.array_equal(cnt, ref)
        assert ref.mean() > 0.5 * len(p1)


def test_score_valid_flags_and_empty(dev, fund_models):
    sc, p1, p2, models = fund_models
    pts4 = np.concatenate([p1, p2], 1).astype(np.float32)
    valid = np.ones(len(models), np.uint8); valid[::3] = 0
    cnt = ops.score_fund(_d(models, dev), _d(pts4, dev), 1.0, _d(valid, dev)).cpu().numpy()
    full = ops.score_fund(_d(models, dev), _d(pts4, dev), 1.0).cpu().numpy()
    assert np.all(cnt[::3] == -1) and np.array_equal(cnt[valid == 1], full[valid == 1])
    assert ops.score_fund(_d(models, dev), _d(pts4[:0], dev), 1.0).cpu().numpy().sum() == 0
    assert ops.score_fund(_d(models[:0], dev), _d(pts4, dev), 1.0).numel() == 0


@pytest.mark.parametrize("thr", [1.0, 5.0])
def test_score_and_mask_pnp(oracle_mod, dev, thr):
    ps = synth.pnp_scene(3000, seed=12)
    obj, img, K = ps["obj"], ps["img"], ps["K"]
    subs = oracle_mod.gen_subsets(len(obj), 4, 500)
    pm = []
    for s in subs:
        ns, rv, tv, R = oracle_mod.p3p_kernel(obj[s], img[s], K)
        if ns > 0:
            pm.append(np.concatenate([R.reshape(9), tv]))
    pm = np.array(pm)
    cnt = ops.score_pnp(_d(pm, dev), _d(obj, dev), _d(img, dev), K, thr).cpu().numpy()
    ref = np.array([oracle_mod.find_inliers(oracle_mod.pnp_error(m[:9], m[9:], K, obj, img), thr)[0] for m in pm])
    assert np.array_equal(cnt, ref)
    b = int(np.argmax(ref))
    m = ops.mask_pnp(_d(pm[b], dev), _d(obj, dev), _d(img, dev), K, thr).cpu().numpy()
    assert np.array_equal(m, oracle_mod.find_inliers(oracle_mod.pnp_error(pm[b][:9], pm[b][9:], K, obj, img), thr)[1])


def _pnp_models(o, obj, img, K, n_subs=300, seed=0):
    """P3P solutions of random minimal samples (near-true and junk poses alike), as (H,12) [R|t] rows."""
    pm = []
    for s in o.gen_subsets(len(obj), 4, n_subs):
        ns, rv, tv, R = o.p3p_kernel(obj[s], img[s], K)
        if ns > 0:
            pm.append(np.concatenate([R.reshape(9), tv]))
    return np.array(pm)


def _pnp_ref(o, models, K, obj, img, thr):
    with np.errstate(all="ignore"):
        return np.array([o.find_inliers(o.pnp_error(m[:9].reshape(3, 3), m[9:], K, obj, img), thr)[0] for m in models])


def test_score_pnp_filter_at_every_boundary(oracle_mod, dev):
    """The PnP count kernel decides most correspondences with a division-free f32 filter and takes the rest
    through the exact f64 evaluation: put the threshold ON the reprojection error of many different points
    (and one f32 ulp either side), for the best and for a mediocre pose."""
    ps = synth.pnp_scene(1500, seed=21)
    obj, img, K = ps["obj"], ps["img"], ps["K"]
    pm = _pnp_models(oracle_mod, obj, img, K, 200)
    counts = _pnp_ref(oracle_mod, pm, K, obj, img, 2.0)
    checked = 0
    for m in (pm[int(np.argmax(counts))], pm[len(pm) // 2]):
        err = oracle_mod.pnp_error(m[:9].reshape(3, 3), m[9:], K, obj, img)
        for e in np.sort(err)[5:1400:31]:
            for t2 in (e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(np.inf))):
                thr = float(np.sqrt(np.float64(t2)))
                if np.float32(thr * thr) != t2 or not np.isfinite(t2):
                    continue
                cnt = ops.score_pnp(_d(m[None], dev), _d(obj, dev), _d(img, dev), K, thr).cpu().numpy()
                assert cnt[0] == int((err <= t2).sum()), (thr, cnt[0])
                checked += 1
    assert checked > 40


def test_score_pnp_filter_degenerate_and_scaled(oracle_mod, dev):
    """Zero / NaN / Inf / absurdly scaled poses, poses whose third row vanishes (z == 0: OpenCV then projects
    with z = 1), points on the camera plane, far-away and non-finite points: every such case must come out of
    the exact evaluation (same counts as the scalar definition)."""
    ps = synth.pnp_scene(700, seed=22)
    obj, img, K = ps["obj"].copy(), ps["img"].copy(), ps["K"]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    base = _pnp_models(oracle_mod, obj, img, K, 40)[:24]
    extra = [np.zeros(12), np.full(12, np.nan), np.full(12, np.inf)]
    for k in range(12):
        m = base[k].copy(); m[k] = np.nan; extra.append(m)
        m = base[k].copy(); m[k] = np.inf; extra.append(m)
        m = base[k].copy(); m[k] = 0.0; extra.append(m)
    # z == 0 for every point: x = X, y = Y, so points with X = (u - cx)/fx, Y = (v - cy)/fy are exact inliers
    m = np.zeros(12); m[0] = 1.0; m[4] = 1.0; extra.append(m)
    m = base[0].copy(); m[6:9] = 0.0; m[11] = 0.0; extra.append(m)
    # z tiny but not zero, and z == 0 only on the plane Z = 0
    m = base[1].copy(); m[6:9] *= 1e-12; m[11] *= 1e-12; extra.append(m)
    m = np.zeros(12); m[0] = 1.0; m[4] = 1.0; m[8] = 1.0; extra.append(m)
    scaled = [base[i] * s_ for i, s_ in enumerate([1e-300, 1e-160, 1e-100, 1e-30, 1e30, 1e100, 1e160, 1e300])]
    mods = np.concatenate([base, np.array(extra), np.array(scaled)])
    obj[40:60, 0] = (img[40:60, 0] - cx) / fx; obj[40:60, 1] = (img[40:60, 1] - cy) / fy     # inliers of the z == 0 poses
    obj[60:80, 2] = 0.0                                                                         # on the plane Z = 0
    seen, z0 = 0, len(base) + 3 + 36                          # z0: index of the first z == 0 pose
    for scale, poison in ((1.0, False), (1.0, True), (1e3, False), (1e-3, False)):
        o2 = (obj.astype(np.float64) * scale).astype(np.float32); i2 = img.copy()
        if poison:
            o2[3, 0] = np.nan; i2[10, 1] = np.inf; o2[20] = 0; i2[21, 0] = 3e38; o2[22, 2] = 3e38; o2[23, 1] = -np.inf
        for thr in (1.0, 5.0, 1e-3):
            ref = _pnp_ref(oracle_mod, mods, K, o2, i2, thr)
            cnt = ops.score_pnp(_d(mods, dev), _d(o2, dev), _d(i2, dev), K, thr).cpu().numpy()
            assert np.array_equal(cnt, ref), (scale, poison, thr, np.nonzero(cnt != ref)[0][:10])
            seen = max(seen, int(ref.max()))
            if scale == 1.0 and not poison and thr == 1.0:
                assert ref[z0] >= 15                              # the z == 0 pose really has inliers (z := 1 branch)
    assert seen > 100


@pytest.mark.parametrize("seed", range(6))
def test_score_pnp_fuzz(oracle_mod, dev, seed):
    """random scenes, scene scales, intrinsics, pose perturbations and thresholds (0.01 px to 30 px): counts equal
    the scalar definition whatever tier decides."""
    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(33, 2500))
    K = [synth.K_MALAGA, synth.K_KITTI, synth.K_PARKING][seed % 3]
    size = [synth.IMAGE_SIZE["malaga"], synth.IMAGE_SIZE["kitti"], synth.IMAGE_SIZE["parking"]][seed % 3]
    ps = synth.pnp_scene(n, K=K, size=size, seed=400 + seed, outlier_frac=float(rng.uniform(0, 0.6)), noise_px=float(rng.uniform(0, 2)))
    scale = float(10 ** rng.uniform(-2, 2))                     # metres -> other units: t scales with the points
    obj = (ps["obj"].astype(np.float64) * scale).astype(np.float32); img = ps["img"]
    pm = _pnp_models(oracle_mod, obj, img, K, 120)
    jit = pm * (1 + rng.normal(0, 1e-3, pm.shape))              # slightly non-orthogonal, slightly wrong poses
    true = np.concatenate([ps["R"].reshape(9), ps["t"] * scale])[None]
    models = np.concatenate([pm, jit, true])
    best = 0
    for thr in list(10 ** rng.uniform(-2, 1.5, 4)) + [8.0]:
        cnt = ops.score_pnp(_d(models, dev), _d(obj, dev), _d(img, dev), K, float(thr)).cpu().numpy()
        ref = _pnp_ref(oracle_mod, models, K, obj, img, float(thr))
        assert np.array_equal(cnt, ref), (seed, thr, np.nonzero(cnt != ref)[0][:5])
        best = max(best, int(ref[-1]))
    assert best > 0.2 * n                                       # the true pose collects the scene's inliers at 8 px


def test_score_and_mask_ess(oracle_mod, dev):
    sc = synth.two_view_scene(3000, seed=11)
    p1, p2, K = sc["p1"], sc["p2"], sc["K"]
    Kinv = np.linalg.inv(K)
    x1 = (np.c_[p1.astype(np.float64), np.ones(len(p1))] @ Kinv.T)[:, :2]
    x2 = (np.c_[p2.astype(np.float64), np.ones(len(p2))] @ Kinv.T)[:, :2]
    subs = oracle_mod.gen_subsets(len(p1), 5, 100)
    em = np.array([E.reshape(9) for s in subs for E in oracle_mod.run5point(x1[s], x2[s])])
    x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
    thr = 1.0 / K[0, 0]
    cnt = ops.score_ess(_d(em, dev), _d(x4, dev), thr).cpu().numpy()
    ref = np.array([oracle_mod.find_inliers(oracle_mod.sampson_error(m, x1, x2), thr)[0] for m in em])
    assert np.array_equal(cnt, ref)
    b = int(np.argmax(ref))
    m = ops.mask_ess(_d(em[b], dev), _d(x4, dev), thr).cpu().numpy()
    assert np.array_equal(m, oracle_mod.find_inliers(oracle_mod.sampson_error(em[b], x1, x2), thr)[1])


def _ess_scene(o, n, seed, n_subs=80, K=None, outlier_frac=0.3, noise_px=0.3):
    """K-normalised f64 correspondences of a two-view scene and five-point solutions of random samples."""
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=outlier_frac, noise_px=noise_px) if K is None else \
        synth.two_view_scene(n, K=K, seed=seed, outlier_frac=outlier_frac, noise_px=noise_px)
    Kinv = np.linalg.inv(sc["K"])
    x1 = (np.c_[sc["p1"].astype(np.float64), np.ones(n)] @ Kinv.T)[:, :2]
    x2 = (np.c_[sc["p2"].astype(np.float64), np.ones(n)] @ Kinv.T)[:, :2]
    em = np.array([E.reshape(9) for s in o.gen_subsets(n, 5, n_subs) for E in o.run5point(x1[s], x2[s])])
    return sc, np.ascontiguousarray(x1), np.ascontiguousarray(x2), em


def _ess_ref(o, models, x1, x2, thr):
    with np.errstate(all="ignore"):
        return np.array([o.find_inliers(o.sampson_error(m, x1, x2), thr)[0] for m in models])


def test_score_ess_filter_at_every_boundary(oracle_mod, dev):
    """The E count kernel decides most correspondences with the f32 filter on rounded coordinates and takes the rest
    through the exact f64 Sampson distance: put the threshold ON the error of many different points (and one f32 ulp
    either side), for the best and for a mediocre model."""
    sc, x1, x2, em = _ess_scene(oracle_mod, 1500, 31)
    x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
    counts = _ess_ref(oracle_mod, em, x1, x2, 1.0 / sc["K"][0, 0])
    checked = 0
    for m in (em[int(np.argmax(counts))], em[len(em) // 2]):
        err = oracle_mod.sampson_error(m, x1, x2)
        for e in np.sort(err)[5:1400:31]:
            for t2 in (e, np.nextafter(e, np.float32(0)), np.nextafter(e, np.float32(np.inf))):
                thr = float(np.sqrt(np.float64(t2)))
                if np.float32(thr * thr) != t2 or not np.isfinite(t2) or t2 <= 0:
                    continue
                cnt = ops.score_ess(_d(m[None], dev), _d(x4, dev), thr).cpu().numpy()
                assert cnt[0] == int((err <= t2).sum()), (thr, cnt[0])
                checked += 1
    assert checked > 40


def test_score_ess_filter_degenerate_and_scaled(oracle_mod, dev):
    """Zero / NaN / Inf / absurdly scaled models, zero denominators, far-away, tiny and non-finite coordinates (also
    values f32 cannot hold): every such case must come out of the exact evaluation."""
    sc, x1, x2, em = _ess_scene(oracle_mod, 700, 32)
    base = em[:24].copy()
    extra = [np.zeros(9), np.full(9, np.nan), np.full(9, np.inf)]
    for k in range(9):
        m = base[k].copy(); m[k] = np.nan; extra.append(m)
        m = base[k].copy(); m[k] = np.inf; extra.append(m)
        m = base[k].copy(); m[k] = 0.0; extra.append(m)
    m = np.zeros(9); m[8] = 1.0; extra.append(m)                      # all four gradient terms vanish
    m = np.zeros(9); m[2] = 1.0; m[6] = -1.0; extra.append(m)
    scaled = [base[i] * s_ for i, s_ in enumerate([1e-300, 1e-160, 1e-150, 1e-100, 1e100, 1e150, 1e160, 1e300])]
    mods = np.concatenate([base, np.array(extra), np.array(scaled)])
    thr0 = 1.0 / sc["K"][0, 0]
    seen = 0
    for scale, poison in ((1.0, False), (1.0, True), (1e4, False), (1e-3, False)):
        q1, q2 = x1 * scale, x2 * scale
        if poison:
            q1[3, 0] = np.nan; q2[10, 1] = np.inf; q1[20] = 0; q2[20] = 0; q1[21, 0] = 1e300; q2[22, 1] = 1e-320
            q1[23, 1] = 3e38; q2[24, 0] = -1e39
        x4 = np.ascontiguousarray(np.concatenate([q1, q2], 1))
        for thr in (thr0 * scale, 20 * thr0 * scale, 1e-3 * thr0 * scale):
            ref = _ess_ref(oracle_mod, mods, q1, q2, thr)
            cnt = ops.score_ess(_d(mods, dev), _d(x4, dev), thr).cpu().numpy()
            assert np.array_equal(cnt, ref), (scale, poison, thr, np.nonzero(cnt != ref)[0][:10])
            seen = max(seen, int(ref.max()))
    assert seen > 100


@pytest.mark.parametrize("seed", range(6))
def test_score_ess_fuzz(oracle_mod, dev, seed):
    """random scenes, intrinsics, model scalings and thresholds (0.01 px to 30 px over the focal length): counts equal
    the scalar definition whatever tier decides."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(33, 2500))
    K = [synth.K_MALAGA, synth.K_KITTI, synth.K_PARKING][seed % 3]
    sc, x1, x2, em = _ess_scene(oracle_mod, n, 600 + seed, 60, K=K, outlier_frac=float(rng.uniform(0, 0.6)),
                                noise_px=float(rng.uniform(0, 2)))
    models = em * (10 ** rng.uniform(-30, 30, (len(em), 1)))          # E is defined up to scale
    jit = em * (1 + rng.normal(0, 1e-3, em.shape))
    models = np.concatenate([models, jit])
    x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
    best = 0
    for thr in list(10 ** rng.uniform(-2, 1.5, 4) / K[0, 0]) + [3.0 / K[0, 0]]:
        cnt = ops.score_ess(_d(models, dev), _d(x4, dev), float(thr)).cpu().numpy()
        ref = _ess_ref(oracle_mod, models, x1, x2, float(thr))
        assert np.array_equal(cnt, ref), (seed, thr, np.nonzero(cnt != ref)[0][:5])
        best = max(best, int(ref.max()))
    assert best > 0.2 * n


def test_replay_and_argmax(oracle_mod, dev):
    rng = np.random.default_rng(0)
    for trial in range(6):
        cnts = rng.integers(0, 50, size=(1000, 3)).astype(np.int32)
        nm = rng.integers(0, 4, size=1000).astype(np.int32)
        if trial % 2 == 0:
            cnts[200, 1] = 900; nm[200] = 3
            cnts[700, 0] = 2500; nm[700] = 3
        if trial == 4:
            cnts[:] = 3                  # never beats modelPoints-1 = 6
        if trial == 5:
            cnts[10:20, :] = 400; nm[10:20] = 3      # plateau: the first one wins
        res = ops.ransac_replay(_d(cnts, dev), _d(nm, dev), 3000, 7, 0.9999, 1000).cpu().numpy()
        assert tuple(int(v) for v in res) == oracle_mod.ransac_replay(cnts, nm, 3000, 7, 0.9999, 1000)
        am = ops.argmax_first(_d(cnts.reshape(-1), dev)).cpu().numpy()
        flat = cnts.reshape(-1)
        assert am[0] == flat.max() and am[1] == int(np.argmax(flat))


def test_solvers_bit_exact_in_det_math(det_math, dev):
    o = det_math
    sc = synth.two_view_scene(2000, seed=21)
    p1, p2 = sc["p1"], sc["p2"]
    pts4 = np.concatenate([p1, p2], 1)
    subs = o.gen_subsets(len(p1), 7, 512, p1, p2)
    assert np.array_equal(subs, ops.gen_subsets(len(p1), 7, 512, pts4))
    models, nm = ops.solve_7pt(_d(subs, dev), _d(pts4, dev))
    models, nm = models.cpu().numpy(), nm.cpu().numpy()
    for h, s in enumerate(subs):
        Fs = o.run7point(p1[s], p2[s])
        assert len(Fs) == nm[h]
        for k in range(nm[h]):
            assert np.array_equal(models[h, k].reshape(3, 3), Fs[k])

    ps = synth.pnp_scene(2000, seed=22)
    obj, img, K = ps["obj"], ps["img"], ps["K"]
    subs = o.gen_subsets(len(obj), 4, 512)
    pm, rv, nm = ops.solve_p3p(_d(subs, dev), _d(obj, dev), _d(img, dev), K)
    pm, rv, nm = pm.cpu().numpy(), rv.cpu().numpy(), nm.cpu().numpy()
    for h, s in enumerate(subs):
        ns, r, t, R = o.p3p_kernel(obj[s], img[s], K)
        assert (ns > 0) == (nm[h] > 0)
        if ns > 0:
            assert np.array_equal(pm[h], np.concatenate([R.reshape(9), t])) and np.array_equal(rv[h], r)

    Kinv = np.linalg.inv(sc["K"])
    x1 = (np.c_[p1.astype(np.float64), np.ones(len(p1))] @ Kinv.T)[:, :2]
    x2 = (np.c_[p2.astype(np.float64), np.ones(len(p2))] @ Kinv.T)[:, :2]
    x4 = np.ascontiguousarray(np.concatenate([x1, x2], 1))
    subs = o.gen_subsets(len(p1), 5, 256)
    em, nm = ops.solve_5pt(_d(subs, dev), _d(x4, dev))
    em, nm = em.cpu().numpy(), nm.cpu().numpy()
    for h, s in enumerate(subs):
        Es = o.run5point(x1[s], x2[s])
        assert len(Es) == nm[h]
        for k in range(nm[h]):
            assert np.array_equal(em[h, k].reshape(3, 3), Es[k])


# ---------------------------------------------------------------- end to end, cv2-shaped
@pytest.mark.parametrize("n,thr,conf,iters,seed", [(2000, 0.1, 0.9999, 100, 1), (600, 1.0, 0.999, 300, 2),
                                                  (4096, 0.1, 0.9999, 2000, 3), (15, 3.0, 0.99, 50, 4)])
def test_find_fundamental_end_to_end(det_math, dev, n, thr, conf, iters, seed):
    sc = synth.two_view_scene(n, seed=seed)
    F, mask, info = ransac.find_fundamental_ransac(sc["p1"].astype(np.float64), sc["p2"].astype(np.float64), thr, conf, iters)
    oF, omask, oinfo = det_math.find_fundamental_mat(sc["p1"], sc["p2"], thr, conf, iters)
    assert (info.best_iter, info.best_model, info.max_good, info.niters) == tuple(int(v) for v in oinfo)
    assert (F is None) == (oF is None)
    assert np.array_equal(mask, omask) and mask.shape == (n, 1) and mask.dtype == np.uint8
    if F is not None:
        assert np.array_equal(F, oF)


@pytest.mark.parametrize("n", [8, 9, 11, 14])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_find_fundamental_lmeds_8_to_14_points(det_math, dev, n, seed):
    """8..14 points: OpenCV's findFundamentalMat takes the LMedS registrator; F, mask and the
    winning hypothesis equal the oracle's restatement of it."""
    o = det_math
    sc = synth.two_view_scene(n, seed=900 + 7 * n + seed, outlier_frac=0.2 if seed else 0.0, noise_px=0.3)
    F, mask, info = ransac.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99, 200)
    oF, omask, oinfo = o.find_fundamental_mat(sc["p1"], sc["p2"], 1.0, 0.99, 200)
    assert (info.best_iter, info.best_model, info.max_good, info.niters) == tuple(int(v) for v in oinfo)
    assert np.array_equal(mask, omask)
    assert (F is None) == (oF is None)
    if F is not None:
        assert np.array_equal(F, oF) and abs(np.linalg.det(F)) < 1e-6 * np.abs(F).max() ** 3 + 1e-12
    # and through the reference-shaped entry points
    F2, mask2 = geometry.calc_fundamental_mat(sc["p1"], sc["p2"])
    F3, mask3 = cv2_shim.findFundamentalMat(sc["p1"], sc["p2"], cv2_shim.RANSAC, 1.0, 0.99, 200)
    assert np.array_equal(mask3, omask) and (F3 is None) == (oF is None)
    assert mask2.shape == (n, 1)


def test_lmeds_median_kernel_orders_like_int_keys(dev):
    """rank N/2 under the int reinterpretation of the f32 errors, +Inf for models that do not exist, and
    a clear error beyond 64 points."""
    from visual_odometry_amd import _lib
    rng = np.random.default_rng(4)
    sc = synth.two_view_scene(64, seed=31)
    for n in (1, 2, 8, 13, 14, 33, 64):
        pts4 = np.concatenate([sc["p1"][:n], sc["p2"][:n]], 1).astype(np.float32)
        models = rng.normal(size=(40, 9)); models[3] = np.nan; models[5] = 0.0
        valid = np.ones(40, np.uint8); valid[7] = 0
        med = ops.lmeds_median_fund(_d(models, dev), _d(pts4, dev), _d(valid, dev)).cpu().numpy()
        import oracle
        for h in range(40):
            if not valid[h]:
                assert np.isposinf(med[h]); continue
            with np.errstate(all="ignore"):
                e = oracle.fund_error(models[h], pts4[:, :2], pts4[:, 2:])
            want = np.sort(e.view(np.int32))[n // 2]
            assert med[h:h + 1].view(np.int32)[0] == want, (n, h)
    with pytest.raises(_lib.VoError):
        ops.lmeds_median_fund(_d(rng.normal(size=(4, 9)), dev), _d(np.zeros((65, 4), np.float32), dev))


def test_find_fundamental_small_inputs(dev):
    sc = synth.two_view_scene(20, seed=3)
    assert cv2_shim.findFundamentalMat(sc["p1"][:6], sc["p2"][:6], cv2_shim.RANSAC, 1.0, 0.99, 10) == (None, None)
    F, mask = cv2_shim.findFundamentalMat(sc["p1"][:7], sc["p2"][:7], cv2_shim.RANSAC, 1.0, 0.99, 10)
    assert F.shape[1] == 3 and F.shape[0] in (3, 6, 9) and mask.sum() == 7
    F, mask = cv2_shim.findFundamentalMat(sc["p1"][:10], sc["p2"][:10], cv2_shim.RANSAC, 1.0, 0.99, 10)   # LMedS (A.3)
    assert mask.shape == (10, 1) and (F is None or F.shape == (3, 3))


@pytest.mark.parametrize("n,thr,iters,seed", [(1500, 1.0, 200, 1), (300, 0.5, 500, 2)])
def test_find_essential_end_to_end(det_math, dev, n, thr, iters, seed):
    sc = synth.two_view_scene(n, seed=seed)
    E, mask = geometry.calc_essential_mat(sc["p1"], sc["p2"], sc["K"], thr, 0.999, iters)
    oE, omask, oinfo = det_math.find_essential_mat(sc["p1"], sc["p2"], sc["K"], 0.999, thr, iters)
    assert np.array_equal(mask, omask)
    assert np.array_equal(E, oE)


@pytest.mark.parametrize("n,thr,conf,iters,seed", [(2000, 1.0, 0.9999, 100, 1), (500, 5.0, 0.9999, 1000, 2),
                                                  (4096, 1.0, 0.9999, 4096, 3), (5, 8.0, 0.99, 20, 4)])
def test_solve_pnp_ransac_end_to_end(det_math, dev, n, thr, conf, iters, seed):
    ps = synth.pnp_scene(n, seed=seed, outlier_frac=0.3 if n > 10 else 0.0)
    ok, rvec, tvec, inl = cv2_shim.solvePnPRansac(ps["obj"].astype(np.float64), ps["img"], ps["K"], np.zeros((4, 1)),
                                                  iterationsCount=iters, reprojectionError=thr, confidence=conf,
                                                  flags=cv2_shim.SOLVEPNP_P3P)
    ook, orvec, otvec, oinl, oinfo = det_math.solve_pnp_ransac(ps["obj"], ps["img"], ps["K"], iters, thr, conf)
    assert ok == ook
    if ok:
        assert inl.dtype == np.int32 and inl.shape[1] == 1 and np.array_equal(inl, oinl)
        assert np.allclose(rvec, orvec, rtol=1e-6, atol=1e-9) and np.allclose(tvec, otvec, rtol=1e-6, atol=1e-9)
        # the pose is the right one: close to ground truth
        R = cv2_shim.Rodrigues(rvec)[0]
        assert np.abs(R - ps["R"]).max() < 5e-3 and np.abs(tvec.ravel() - ps["t"]).max() < 5e-2


def test_reference_level_functions(det_math, dev):
    """calc_fundamental_mat / pnp with the reference's per-dataset parameters."""
    for ds in (0, 1, 2):
        params.which_dataset(ds)
        cfg = params.get_params()
        sc = synth.two_view_scene(1200, seed=30 + ds)
        F, mask = geometry.calc_fundamental_mat(sc["p1"].astype(np.float64), sc["p2"].astype(np.float64))
        c = cfg.INIT_PARAMS.RANSAC_PARAMS_F
        oF, omask, _ = det_math.find_fundamental_mat(sc["p1"], sc["p2"], c.THRESHOLD, c.CONFIDENCE, c.NUM_ITERATIONS)
        assert np.array_equal(mask, omask) and ((F is None and oF is None) or np.array_equal(F, oF))
        E = geometry.calc_essential_mat_from_fundamental_mat(F, sc["K"])
        assert np.array_equal(E, sc["K"].T @ F @ sc["K"])

        ps = synth.pnp_scene(800, seed=40 + ds)
        T, inl = pnp(ps["img"], ps["obj"].astype(np.float64), ps["K"])
        c = cfg.CONT_PARAMS.PNP_RANSAC_PARAMS
        oT, oinl = det_math.pnp(ps["img"], ps["obj"], ps["K"], c.NUM_ITERATIONS, c.THRESHOLD, c.CONFIDENCE)
        assert np.array_equal(inl, oinl)
        assert np.allclose(T, oT, rtol=1e-6, atol=1e-9)
        assert np.squeeze(inl).ndim == 1          # usable as a fancy index, reference src/tracking.py:410-414
    params.which_dataset(0)


# ---------------------------------------------------------------- fused batched pipeline
def test_replay_long_and_batched(oracle_mod, dev):
    """the parallel replay on tables longer than one 1024-entry step, several problems per launch"""
    rng = np.random.default_rng(3)
    B, n_it, mm = 5, 3000, 3
    cnts = rng.integers(0, 60, size=(B, n_it, mm)).astype(np.int32)
    nm = rng.integers(0, mm + 1, size=(B, n_it)).astype(np.int32)
    cnts[1, 2500, 2] = 900; nm[1, 2500] = 3                    # late record, niters still large
    cnts[2, 10, 0] = 2990; nm[2, 10] = 1                       # early near-perfect model: scan stops
    cnts[3, np.arange(0, 3000, 7), 1] = np.arange(100, 100 + len(np.arange(0, 3000, 7)))  # many records
    nm[3, np.arange(0, 3000, 7)] = 3
    cnts[4] = 2
    from visual_odometry_amd import _lib
    lib = _lib.load()
    for b in range(B):
        res = ops.ransac_replay(_d(cnts[b], dev), _d(nm[b], dev), 3000, 7, 0.9999, n_it).cpu().numpy()
        assert tuple(int(v) for v in res) == oracle_mod.ransac_replay(cnts[b], nm[b], 3000, 7, 0.9999, n_it), b


@pytest.mark.parametrize("B,H,N,first_block", [(3, 512, 700, 0), (1, 100, 2000, 0), (6, 1100, 333, 0),
                                                (3, 512, 700, 64), (6, 1100, 333, 37), (2, 2000, 500, 16)])
def test_ransac_batch_fund_and_pnp(det_math, dev, B, H, N, first_block):
    """first_block > 0: iterations run in blocks with early termination -- results must not change."""
    o = det_math
    f_pts, f_sub, p_obj, p_img, p_sub = [], [], [], [], []
    for b in range(B):
        sc = synth.two_view_scene(N, seed=60 + b)
        pts4 = np.concatenate([sc["p1"], sc["p2"]], 1).astype(np.float32)
        f_pts.append(pts4); f_sub.append(ops.gen_subsets(N, 7, H, pts4))
        ps = synth.pnp_scene(N, seed=80 + b)
        p_obj.append(ps["obj"]); p_img.append(ps["img"]); p_sub.append(ops.gen_subsets(N, 4, H))
    K = synth.K_MALAGA
    res, best, _, mask = ops.ransac_batch(ops.KIND_FUND, _d(np.stack(f_sub), dev), _d(np.stack(f_pts), dev), 1.0, 0.999, H,
                                          first_block=first_block)
    resp, bestp, rvp, maskp = ops.ransac_batch(ops.KIND_PNP, _d(np.stack(p_sub), dev), _d(np.stack(p_obj), dev), 2.0, 0.999, H,
                                               img=_d(np.stack(p_img), dev), K=K, first_block=first_block)
    res, best, mask = res.cpu().numpy(), best.cpu().numpy(), mask.cpu().numpy()
    resp, bestp, rvp, maskp = resp.cpu().numpy(), bestp.cpu().numpy(), rvp.cpu().numpy(), maskp.cpu().numpy()
    for b in range(B):
        F, omask, info = o.find_fundamental_mat(f_pts[b][:, :2], f_pts[b][:, 2:], 1.0, 0.999, H)
        assert tuple(res[b]) == tuple(int(v) for v in info)
        assert np.array_equal(mask[b], omask.ravel()) and np.array_equal(best[b].reshape(3, 3), F)
        # P3P stage of solvePnPRansac: compare decisions and the winning model with a by-hand oracle run
        subs = p_sub[b]
        cnts = np.zeros(H, np.int32); nm = np.zeros(H, np.int32); models = np.zeros((H, 12)); rvs = np.zeros((H, 3))
        for h, s in enumerate(subs):
            ns, rv, tv, R = o.p3p_kernel(p_obj[b][s], p_img[b][s], K)
            if ns > 0:
                nm[h] = 1; models[h] = np.concatenate([R.reshape(9), tv]); rvs[h] = rv
                cnts[h] = o.find_inliers(o.pnp_error(R, tv, K, p_obj[b], p_img[b]), 2.0)[0]
        exp = o.ransac_replay(cnts, nm, N, 4, 0.999, H)
        assert tuple(resp[b]) == exp
        if exp[0] >= 0:
            m = models[exp[0]]
            assert np.array_equal(bestp[b], m) and np.array_equal(rvp[b], rvs[exp[0]])
            assert np.array_equal(maskp[b], o.find_inliers(o.pnp_error(m[:9], m[9:], K, p_obj[b], p_img[b]), 2.0)[1])


def test_ransac_batch_ess_and_n_dev(det_math, dev):
    o = det_math
    B, H, N = 2, 150, 400
    sc = [synth.two_view_scene(N, seed=90 + b) for b in range(B)]
    K = sc[0]["K"]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x4s, subs = [], []
    nvalid = [N, 250]                 # the second problem only uses its first 250 correspondences
    for b in range(B):
        p1, p2 = sc[b]["p1"].astype(np.float64), sc[b]["p2"].astype(np.float64)
        x4 = np.stack([p1[:, 0] * (1 / fx) + (-cx * (1 / fx)), p1[:, 1] * (1 / fy) + (-cy * (1 / fy)),
                       p2[:, 0] * (1 / fx) + (-cx * (1 / fx)), p2[:, 1] * (1 / fy) + (-cy * (1 / fy))], 1)
        x4s.append(x4)
        s = np.zeros((H, 5), np.int32); g = ops.gen_subsets(nvalid[b], 5, H); s[:len(g)] = g
        subs.append(s)
    thr = 1.0 / ((fx + fy) / 2)
    res, best, _, mask = ops.ransac_batch(ops.KIND_ESS, _d(np.stack(subs), dev), _d(np.stack(x4s), dev), thr, 0.999, H,
                                          n_dev=_d(np.array(nvalid, np.int32), dev), first_block=20)
    res, best, mask = res.cpu().numpy(), best.cpu().numpy(), mask.cpu().numpy()
    for b in range(B):
        n = nvalid[b]
        E, omask, info = o.find_essential_mat(sc[b]["p1"][:n], sc[b]["p2"][:n], K, 0.999, 1.0, H)
        assert tuple(res[b]) == tuple(int(v) for v in info)
        assert np.array_equal(best[b].reshape(3, 3), E)
        assert np.array_equal(mask[b][:n], omask.ravel()) and not mask[b][n:].any()
