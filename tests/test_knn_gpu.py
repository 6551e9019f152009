"""GPU parity: descriptor matching (K1) vs the oracle's BFMatcher restatement, bit-exact."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import _lib, features, ops, synth

pytestmark = pytest.mark.gpu


def _gpu_knn(q, t, dev, workspace=None):
    qi8, qn, f1 = ops.desc_to_i8(torch.from_numpy(q).to(dev))
    ti8, tn, f2 = ops.desc_to_i8(torch.from_numpy(t).to(dev))
    idx, d2 = ops.knn2(qi8, qn, ti8, tn, workspace)
    return idx.cpu().numpy(), d2.cpu().numpy(), int(f1.item()) | int(f2.item())


def _with_ties(q, t):
    if q.shape[0] >= 64 and t.shape[0] >= 64:
        t[5] = t[3]; t[40] = t[3]; q[7] = t[3]          # exact duplicates: lower index must win
        q[9] = 0                                         # a zero row
        t[33] = t[2]; t[34] = t[2]                       # duplicates inside one 32-row tile
    return q, t


@pytest.mark.parametrize("nq,nt,seed", [(64, 64, 1), (100, 37, 2), (1, 2, 3), (3, 1, 4), (300, 1000, 5),
                                        (257, 513, 6), (1024, 1024, 7), (4096, 4096, 8), (8192, 8192, 9),
                                        (640, 20000, 10)])
def test_knn2_matches_oracle(oracle_mod, dev, nq, nt, seed):
    q, t = _with_ties(*synth.sift_like_pair(nq, nt, seed))
    idx, d2, flags = _gpu_knn(q, t, dev)
    oi, od = oracle_mod.knn2(q, t, mt=True)
    assert flags & _lib.VO_DESC_BAD_VALUE == 0
    assert np.array_equal(idx, oi)
    dist = np.sqrt(d2.astype(np.float32))
    dist[idx < 0] = np.finfo(np.float32).max
    assert np.array_equal(dist, od)


def test_knn2_batched_equals_single(oracle_mod, dev):
    B, nq, nt = 3, 500, 700
    qs, ts = zip(*[synth.sift_like_pair(nq, nt, 40 + b) for b in range(B)])
    q = torch.from_numpy(np.stack(qs)).to(dev); t = torch.from_numpy(np.stack(ts)).to(dev)
    qi8, qn, _ = ops.desc_to_i8(q.reshape(-1, 128)); ti8, tn, _ = ops.desc_to_i8(t.reshape(-1, 128))
    idx, d2 = ops.knn2(qi8.reshape(B, nq, 128), qn.reshape(B, nq), ti8.reshape(B, nt, 128), tn.reshape(B, nt))
    for b in range(B):
        oi, od = oracle_mod.knn2(qs[b], ts[b])
        assert np.array_equal(idx[b].cpu().numpy(), oi)
        assert np.array_equal(np.sqrt(d2[b].cpu().numpy().astype(np.float32)), od)


@pytest.mark.parametrize("B,nq,nt", [(8, 600, 2100), (16, 1100, 900), (24, 300, 300)])
def test_knn2_batch_multiple_of_8_distinct_problems(oracle_mod, dev, B, nq, nt):
    """batches that are a multiple of 8 take the XCD-aware re-labelling of the workgroups (every problem
    pinned to one XCD): distinct problems, several query blocks and train splits, each against the oracle."""
    qs, ts = zip(*[synth.sift_like_pair(nq, nt, 300 + 17 * b + B) for b in range(B)])
    q = torch.from_numpy(np.stack(qs)).to(dev); t = torch.from_numpy(np.stack(ts)).to(dev)
    qi8, qn, _ = ops.desc_to_i8(q.reshape(-1, 128)); ti8, tn, _ = ops.desc_to_i8(t.reshape(-1, 128))
    idx, d2 = ops.knn2(qi8.reshape(B, nq, 128), qn.reshape(B, nq), ti8.reshape(B, nt, 128), tn.reshape(B, nt))
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for b in range(B):
        oi, od = oracle_mod.knn2(qs[b], ts[b], mt=True)
        assert np.array_equal(idx[b], oi), b
        assert np.array_equal(np.sqrt(d2[b].astype(np.float32)), od), b


def test_knn2_single_split_path(oracle_mod, dev):
    """batch large enough that one train split per workgroup is used (no merge kernel)."""
    B, nq, nt = 40, 600, 300
    q0, t0 = synth.sift_like_pair(nq, nt, 77)
    q = torch.from_numpy(np.stack([q0] * B)).to(dev); t = torch.from_numpy(np.stack([t0] * B)).to(dev)
    qi8, qn, _ = ops.desc_to_i8(q.reshape(-1, 128)); ti8, tn, _ = ops.desc_to_i8(t.reshape(-1, 128))
    idx, d2 = ops.knn2(qi8.reshape(B, nq, 128), qn.reshape(B, nq), ti8.reshape(B, nt, 128), tn.reshape(B, nt))
    oi, od = oracle_mod.knn2(q0, t0)
    for b in (0, B // 2, B - 1):
        assert np.array_equal(idx[b].cpu().numpy(), oi)
        assert np.array_equal(np.sqrt(d2[b].cpu().numpy().astype(np.float32)), od)


def test_extreme_rows_flagged_and_exact(oracle_mod, dev):
    """all-0 vs all-255 rows: D = 128*255^2 > 2^22, where OpenCV's f32 sqrt can collide; the
    converter must flag the norm, and the integer distances themselves stay exact."""
    q = np.zeros((64, 128), np.float32); t = np.full((64, 128), 255, np.float32)
    t[10, :5] = 254
    idx, d2, flags = _gpu_knn(q, t, dev)
    assert flags & _lib.VO_DESC_BAD_NORM
    assert d2[0, 0] == 128 * 255 * 255 - 5 * (255 * 255 - 254 * 254) and idx[0, 0] == 10
    assert d2[0, 1] == 128 * 255 * 255 and idx[0, 1] == 0


def test_non_integer_descriptors_rejected(dev):
    q, t = synth.sift_like_pair(64, 64, 1)
    q[3, 7] = 0.5
    with pytest.raises(_lib.VoError):
        features.knn2_arrays(q, t)


@pytest.mark.parametrize("ratio", [0.6, 0.8])
def test_ratio_test_device_equals_reference_expression(oracle_mod, dev, ratio):
    q, t = synth.sift_like_pair(2000, 2000, 11)
    qi8, qn, _ = ops.desc_to_i8(torch.from_numpy(q).to(dev)); ti8, tn, _ = ops.desc_to_i8(torch.from_numpy(t).to(dev))
    idx, d2 = ops.knn2(qi8, qn, ti8, tn)
    keep, count = ops.ratio_test(d2, ratio)
    oi, od = oracle_mod.knn2(q, t)
    ref = oracle_mod.ratio_test(od, ratio)
    assert np.array_equal(keep.cpu().numpy().astype(bool), ref)
    assert int(count.item()) == int(ref.sum())


def test_match_features_dropin(oracle_mod, dev, capsys):
    from visual_odometry_amd import params
    q, t = synth.sift_like_pair(900, 1100, 21)
    kq = np.empty(900, dtype=object)          # 1-D object array of keypoints
    for i in range(900):
        kq[i] = features.Keypoint(np.array([float(i), 2.0 * i]), 1.0, 0.0, 1.0, 0, -1)
    kt = [features.Keypoint(np.array([3.0 * i, float(i)]), 1.0, 0.0, 1.0, 0, -1) for i in range(1100)]
    for ds, thr in ((0, 0.8), (1, 0.6)):
        params.which_dataset(ds)
        fi, fj = features.match_features(features.Features(kq, q), features.Features(kt, t))
        oq, ot, _, _ = oracle_mod.match_features_indices(q, t, thr)
        assert f"#SIFT Matches filtered: {len(oq)}" in capsys.readouterr().out
        assert np.array_equal(fi.descriptors, q[oq]) and np.array_equal(fj.descriptors, t[ot])
        assert np.array_equal(fi.get_positions()[:, 0], oq.astype(np.float64))
        assert np.array_equal(fj.get_positions()[:, 1], ot.astype(np.float64))
    params.which_dataset(0)


def test_bfmatcher_shim(oracle_mod, dev):
    from visual_odometry_amd import cv2_shim as cv
    q, t = synth.sift_like_pair(50, 80, 5)
    matches = cv.BFMatcher().knnMatch(q, t, k=2)
    oi, od = oracle_mod.knn2(q, t)
    assert len(matches) == 50
    for i, (m, n) in enumerate(matches):
        assert (m.queryIdx, m.trainIdx, m.imgIdx, n.trainIdx) == (i, oi[i, 0], 0, oi[i, 1])
        assert m.distance == float(od[i, 0]) and n.distance == float(od[i, 1])
    with pytest.raises(ValueError):
        features.match_features_indices(q, t[:1], 0.8)
