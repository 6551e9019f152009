"""The sweep's batched pair generator (host logic; runs on the CPU device here): counter-based draws, so a pair's
data is the same whatever batch it is generated in, and the distributions are the ones synth.py documents."""
import numpy as np
import torch

from visual_odometry_amd import sweep, synth

CPU = torch.device("cpu")


def test_pair_rng_distributions_and_independence():
    r = sweep.PairRng([5, 6, 7], 1234, CPU)
    u = r.uniform(20000).numpy()
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    z = r.normal(20000).numpy()
    assert abs(z.mean()) < 2e-2 and abs(z.var() - 1) < 3e-2
    assert abs(np.corrcoef(u[0], u[1])[0, 1]) < 2e-2 and abs(np.corrcoef(u[0, :-1], u[0, 1:])[0, 1]) < 2e-2
    g = r.gamma06(40000).numpy()
    assert g.min() >= 0 and abs(g.mean() - 0.6) < 0.03 and abs(g.var() - 0.6) < 0.08       # Gamma(k): mean k, variance k
    p = r.permutation(257).numpy()
    assert all(np.array_equal(np.sort(row), np.arange(257)) for row in p) and not np.array_equal(p[0], p[1])
    # the same (seed, pair, stream, index) gives the same draw, alone or in a batch
    a, b = sweep.PairRng([6], 1234, CPU), sweep.PairRng([9, 6], 1234, CPU)
    assert np.array_equal(a.uniform(100).numpy()[0], b.uniform(100).numpy()[1])
    assert not np.array_equal(sweep.PairRng([6], 1235, CPU).uniform(100).numpy()[0], a.uniform(100).numpy()[0])


def test_batched_pairs_equal_single_pairs_and_look_like_sift():
    K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
    batch = sweep.frame_pairs_device([3, 11, 40], 300, 340, K, size, CPU)
    for j, i in enumerate([3, 11, 40]):
        one = sweep.frame_pair_device(i, 300, 340, K, size, CPU)
        for k in batch:
            assert np.array_equal(batch[k][j].numpy(), one[k].numpy()), k
    q, t = batch["q"].numpy(), batch["t"].numpy()
    assert q.dtype == np.float32 and q.shape == (3, 300, 128) and t.shape == (3, 340, 128)
    assert q.min() >= 0 and q.max() <= 255 and np.array_equal(q, np.rint(q))
    assert 450 < np.linalg.norm(q, axis=2).mean() < 560
    # geometry: the true matches satisfy x2^T F x1 = 0 up to the pixel noise
    R, tv = batch["R"].numpy(), batch["t_vec"].numpy()
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-12) and np.allclose(np.linalg.norm(tv, axis=1), 1)
    X = batch["xyz_q"].numpy().astype(np.float64)
    proj = X / X[:, :, 2:3]
    uv = proj[:, :, :2] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
    assert np.abs(uv - batch["kp_q"].numpy()).max() < 3.0            # 0.3 px noise
    # a train keypoint that is a match sits on the projection of its query's landmark
    Xc = X[0] @ R[0].T + tv[0]
    p2 = Xc[:, :2] / Xc[:, 2:3] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
    d = np.linalg.norm(batch["kp_t"].numpy()[0][:, None, :] - p2[None, :, :], axis=2).min(axis=1)
    assert (d < 3.0).sum() >= int(0.6 * 300) - 5
