"""ONE batch of frame pairs over several ranks (pipeline.match_and_ransac(group=...), what `bench.py --mode shard`
times) against the unsharded pipeline on the same inputs, bit for bit, on the MI355X:
  * two ranks that share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU; the data path -- sliced
    kNN, sliced solve + score, table exchange, replicated replay, winner exchange -- is the one N GPUs run);
  * one rank with a live RCCL ("nccl") process group, through both the sharded path (real one-rank collectives)
    and the weak-mode path with its PnP side stream: the configuration that hung in round 1 when that stream
    had high priority."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(dev):
    from visual_odometry_amd import synth
    B, nq, nt = 3, 1024, 900
    fps = [synth.frame_pair(nq, nt, seed=900 + b, K=synth.K_MALAGA, size=synth.IMAGE_SIZE["malaga"],
                            landmark_outlier_frac=0.5 if b == 1 else 0.0) for b in range(B)]
    up = lambda k: torch.from_numpy(np.stack([f[k] for f in fps])).to(dev)
    return up("q"), up("t"), up("kp_q"), up("kp_t"), up("xyz_q"), synth.K_MALAGA


def _compare(a, b):
    """every device field of two PairBatchResults; rows past n_match are unspecified"""
    bad = []
    n = a.n_match.cpu().numpy()
    if not torch.equal(a.n_match, b.n_match):
        return ["n_match"]
    for name in ("idx", "d2", "f_result", "F", "p_result", "pose", "rvec", "pose_refined", "p_ok", "f_samples", "p_samples"):
        if not torch.equal(getattr(a, name), getattr(b, name)):
            bad.append(name)
    for name in ("match_q", "match_t", "pts4", "f_mask", "p_mask", "obj", "img"):
        for i in range(len(n)):
            if not torch.equal(getattr(a, name)[i, :n[i]], getattr(b, name)[i, :n[i]]):
                bad.append(f"{name}[{i}]")
    return bad


def _worker(rank, world, port, backend, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from visual_odometry_amd import _lib, parallel as par, pipeline
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.require_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q, t, kq, kt, xyz, K = _inputs(dev)
        args = (q, t, kq, kt, 0.8, 1.0, 0.999, 300)
        kw = dict(xyz_q=xyz, K=K, p_threshold=2.0, p_confidence=0.999, p_iters=200)
        ref = pipeline.match_and_ransac(*args, **kw)
        bad = []
        for split, fb in (("pairs", 0), ("pairs", 64), ("iterations", 0)):
            got = pipeline.match_and_ransac_sharded(*args, group=par.WORLD, pnp_split=split, first_block_pnp=fb, **kw)
            torch.cuda.synchronize()
            bad += [f"{split}/{fb}: {x}" for x in _compare(ref, got)]
        if backend == "nccl":                                     # weak mode beside a live RCCL communicator
            for _ in range(3):
                w = pipeline.match_and_ransac(*args, first_block_pnp=64, wait_pnp=False, **kw)
            dist.barrier()
            torch.cuda.synchronize()
            bad += _compare(ref, w)
        ok_models = int((ref.f_result[:, 0] >= 0).sum()) == 3 and int((ref.p_result[:, 0] >= 0).sum()) == 3
        out[rank] = (bad, ok_models, int(ref.n_match.min()))
    finally:
        dist.destroy_process_group()


def _run(world, backend):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def test_sharded_pipeline_two_ranks_on_one_gpu(dev):
    for bad, ok_models, nmin in _run(2, "gloo"):
        assert bad == [] and ok_models and nmin > 100


def test_pipeline_beside_live_rccl_group_of_one_rank(dev):
    (bad, ok_models, nmin), = _run(1, "nccl")
    assert bad == [] and ok_models and nmin > 100
