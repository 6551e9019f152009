"""The sweep's device generators (vo_synth_*): same hash words as the tensor-op path of sweep.py (run here on the
CPU device), so the two agree except where a float transform differs in its last bit before a rounding."""
import numpy as np
import pytest
import torch

from visual_odometry_amd import ops, sweep, synth

pytestmark = pytest.mark.gpu


def test_sift_like_kernel_matches_tensor_path(dev):
    ids, n = [3, 11, 40], 700
    g, c = sweep.PairRng(ids, 1234, dev), sweep.PairRng(ids, 1234, torch.device("cpu"))
    a = sweep.sift_like_device(g, n).cpu().numpy()
    b = sweep.sift_like_device(c, n).numpy()
    assert a.shape == b.shape == (3, n, 128) and a.dtype == np.float32
    assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a, np.rint(a))
    assert (a != b).mean() < 2e-3 and np.abs(a - b).max() <= 1            # last-bit differences ahead of the rounding
    assert 450 < np.linalg.norm(a, axis=2).mean() < 560
    # streams advance alike on both paths, and a pair's rows do not depend on its batch
    assert g.stream == c.stream
    one = sweep.sift_like_device(sweep.PairRng([11], 1234, dev), n).cpu().numpy()
    assert np.array_equal(one[0], a[1])


def test_noisy_copies_kernel(dev):
    K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
    ids = [5, 6]
    p = sweep.frame_pairs_device(ids, 600, 650, K, size, dev)
    q, t = p["q"].cpu().numpy(), p["t"].cpu().numpy()
    # 60 % of min(nq, nt) train rows are noisy copies of query rows: nearest query within the noise radius
    for b in range(2):
        d = np.linalg.norm(t[b][:, None, :] - q[b][None, :, :], axis=2).min(axis=1)
        close = d < 8.0 * np.sqrt(128) * 1.5
        assert abs(int(close.sum()) - int(0.6 * 600)) <= 3
    assert t.min() >= 0 and t.max() <= 255 and np.array_equal(t, np.rint(t))
    again = sweep.frame_pair_device(6, 600, 650, K, size, dev)
    for k in p:
        assert np.array_equal(p[k][1].cpu().numpy(), again[k].cpu().numpy()), k
    # argument checks of the wrapper
    with pytest.raises(Exception):
        ops.synth_noisy_copies(torch.zeros(2, dtype=torch.int64, device=dev), p["q"], torch.zeros((2, 4), dtype=torch.int64, device=dev),
                               torch.zeros((2, 5), dtype=torch.int64, device=dev), 1, 1.0, p["t"])


def test_hash_words_equal_the_tensor_path(dev):
    """vo_synth_hash64 = PairRng.bits of the CPU device, word for word, so the f64 scene draws of both devices agree."""
    ids = [0, 7, 123456, 2 ** 40 + 5]
    g, c = sweep.PairRng(ids, 99, dev), sweep.PairRng(ids, 99, torch.device("cpu"))
    for shape in ((1,), (3,), (1000,), (37, 128)):
        assert np.array_equal(g.bits(*shape).cpu().numpy(), c.bits(*shape).numpy())
    assert np.array_equal(g.uniform(500).cpu().numpy(), c.uniform(500).numpy())
    assert np.array_equal(g.permutation(8192).cpu().numpy(), c.permutation(8192).numpy())
    p = g.permutation(1000).cpu().numpy()
    assert all(np.array_equal(np.sort(row), np.arange(1000)) for row in p)
