"""world_size-2 (and 3) gloo runs on CPU of the multi-GPU layout in visual_odometry_amd.parallel:
the collectives and the replicated replay must give exactly the unsharded answer.  The per-rank
"solve and score" step, a GPU kernel batch in production, is played here by the CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


# ------------------------------------------------------------------ workers (run in child processes)
def _w_gather(rank, world):
    from visual_odometry_amd import parallel as par
    n = 1001
    full = torch.arange(n * 3, dtype=torch.int32).reshape(n, 3)
    lo, hi = par.shard_bounds(n, rank, world)
    got = par.all_gather_rows(full[lo:hi].contiguous(), n, par.WORLD)
    return bool(torch.equal(got, full))


def _w_sweep_tables(rank, world):
    """the sweep's pair dealing and table gather (the per-pair device work is replaced by a formula)."""
    from visual_odometry_amd import parallel as par, sweep
    ids = list(range(5, 42, 3))
    mine = par.round_robin(ids, rank, world)
    tab = np.zeros(len(mine), dtype=sweep.TABLE_DTYPE)
    tab["pair"] = mine
    tab["n_match"] = [7 * i + 1 for i in mine]
    tab["rot_err_deg"] = [0.5 * i for i in mine]
    got = sweep.gather_tables(tab, par.WORLD)
    if rank != 0:
        return got is None
    return bool(list(got["pair"]) == ids and list(got["n_match"]) == [7 * i + 1 for i in ids]
                and np.allclose(got["rot_err_deg"], [0.5 * i for i in ids]))


def _w_best(rank, world):
    from visual_odometry_amd import parallel as par
    rng = np.random.default_rng(0)                       # same vector on every rank
    counts = rng.integers(0, 50, size=(5, 999)).astype(np.int32)
    counts[0, 17] = counts[0, 600] = 77                  # tie across ranks: lower index wins
    lo, hi = par.shard_bounds(999, rank, world)
    loc = torch.from_numpy(counts[:, lo:hi])
    c, i = loc.max(dim=1)
    # torch.max returns some index of the maximum; take the first explicitly
    i = torch.tensor([int(np.argmax(counts[p, lo:hi])) for p in range(5)])
    bc, bi = par.allreduce_best(c.to(torch.int32), (i + lo).to(torch.int32), par.WORLD)
    exp_i = counts.argmax(axis=1)
    return bool(np.array_equal(bc.numpy(), counts.max(axis=1)) and np.array_equal(bi.numpy(), exp_i))


def _fund_problem():
    from visual_odometry_amd import synth
    sc = synth.two_view_scene(400, seed=5)
    return sc["p1"], sc["p2"], 1.0, 0.999, 150


def _w_sharded_fund(rank, world):
    import oracle as o
    from visual_odometry_amd import parallel as par
    p1, p2, thr, conf, iters = _fund_problem()
    subs = o.gen_subsets(len(p1), 7, iters, p1, p2)

    def solve_and_score(lo, hi):
        counts = np.full((hi - lo, 3), -1, np.int32); nm = np.zeros(hi - lo, np.int32); models = np.zeros((hi - lo, 3, 9))
        for r, s in enumerate(subs[lo:hi]):
            Fs = o.run7point(p1[s], p2[s])
            nm[r] = len(Fs)
            for k, F in enumerate(Fs):
                models[r, k] = F.reshape(9)
                counts[r, k] = o.find_inliers(o.fund_error(F, p1, p2), thr)[0]
        return torch.from_numpy(counts), torch.from_numpy(nm), torch.from_numpy(models)

    def replay(counts, nm):
        return o.ransac_replay(counts.numpy(), nm.numpy(), len(p1), 7, conf, iters)

    bi, bk, good, niters, model = par.sharded_ransac(len(subs), 3, 9, solve_and_score, replay, par.WORLD)
    return (bi, bk, good, niters, model.numpy().tolist())


class _OracleBackend:
    """CPU stand-in for parallel.DeviceRansacBackend: the same three steps with the oracle's solvers and error
    functions on host tensors (tables laid out like the device workspace: counts (B,H,width), nmodels (B,H))."""

    def __init__(self, o):
        self.o = o

    def slice_counts(self, kind, samples, pts, threshold, lo, hi, img, K, n_dev, workspace):
        o = self.o
        B, H = samples.shape[0], samples.shape[1]
        width, mw = (3, 9) if kind == 0 else (1, 12)
        counts = torch.zeros((B, H, width), dtype=torch.int32); nm = torch.zeros((B, H), dtype=torch.int32)
        models = np.zeros((B, H, width, mw)); rvecs = np.zeros((B, H, 3))
        for b in range(B):
            n = int(n_dev[b]) if n_dev is not None else pts.shape[1]
            P = pts[b, :n].numpy()
            for h in range(lo, hi):
                s = samples[b, h].numpy()
                if s[0] < 0:
                    continue
                if kind == 0:
                    Fs = o.run7point(P[s, :2], P[s, 2:])
                    nm[b, h] = len(Fs)
                    for k, F in enumerate(Fs):
                        models[b, h, k] = F.reshape(9)
                        counts[b, h, k] = o.find_inliers(o.fund_error(F, P[:, :2], P[:, 2:]), threshold)[0]
                else:
                    I = img[b, :n].numpy()
                    ns, rv, tv, R = o.p3p_kernel(P[s], I[s], K)
                    if ns > 0:
                        nm[b, h] = 1; models[b, h, 0] = np.concatenate([R.reshape(9), tv]); rvecs[b, h] = rv
                        counts[b, h, 0] = o.find_inliers(o.pnp_error(R, tv, K, P, I), threshold)[0]
        return counts, nm, dict(counts=counts, nm=nm, models=models, rvecs=rvecs, kind=kind)

    def slice_finish(self, kind, H, lo, hi, N, batch, confidence, max_iters, state, n_dev):
        result = torch.zeros((batch, 4), dtype=torch.int32)
        mw = 9 if kind == 0 else 12
        best = torch.zeros((batch, mw), dtype=torch.float64)
        rvec = torch.zeros((batch, 3), dtype=torch.float64) if kind == 2 else None
        for b in range(batch):
            n = int(n_dev[b]) if n_dev is not None else N
            r = self.o.ransac_replay(state["counts"][b].numpy(), state["nm"][b].numpy(), n, 7 if kind == 0 else 4, confidence, max_iters)
            result[b] = torch.tensor(r, dtype=torch.int32)
            if lo <= r[0] < hi:
                best[b] = torch.from_numpy(state["models"][b, r[0], r[1]])
                if rvec is not None:
                    rvec[b] = torch.from_numpy(state["rvecs"][b, r[0]])
        return result, best, rvec

    def mask_models(self, kind, best, result, pts, threshold, img, K, n_dev):
        B, N = pts.shape[0], pts.shape[1]
        mask = torch.zeros((B, N), dtype=torch.uint8)
        for b in range(B):
            if int(result[b, 0]) < 0:
                continue
            n = int(n_dev[b]) if n_dev is not None else N
            P = pts[b, :n].numpy(); m = best[b].numpy()
            if kind == 0:
                e = self.o.fund_error(m.reshape(3, 3), P[:, :2], P[:, 2:])
            else:
                e = self.o.pnp_error(m[:9].reshape(3, 3), m[9:], K, P, img[b, :n].numpy())
            mask[b, :n] = torch.from_numpy(self.o.find_inliers(e, threshold)[1].astype(np.uint8))
        return mask


def _batch_problems():
    """three F problems and three PnP problems of ragged size (the second uses only its first 180 rows)."""
    import oracle as o
    from visual_odometry_amd import synth
    B, N, H = 3, 260, 90
    nd = np.array([N, 180, N - 1], np.int32)
    f_pts = np.zeros((B, N, 4), np.float32); f_sub = np.full((B, H, 7), -1, np.int32)
    p_obj = np.zeros((B, N, 3), np.float32); p_img = np.zeros((B, N, 2), np.float32); p_sub = np.full((B, H, 4), -1, np.int32)
    for b in range(B):
        sc = synth.two_view_scene(N, seed=500 + b)
        f_pts[b] = np.concatenate([sc["p1"], sc["p2"]], 1)
        g = o.gen_subsets(int(nd[b]), 7, H, f_pts[b, :nd[b], :2], f_pts[b, :nd[b], 2:]); f_sub[b, :len(g)] = g
        ps = synth.pnp_scene(N, seed=600 + b)
        p_obj[b], p_img[b] = ps["obj"], ps["img"]
        g = o.gen_subsets(int(nd[b]), 4, H); p_sub[b, :len(g)] = g
    return dict(B=B, N=N, H=H, nd=nd, f_pts=f_pts, f_sub=f_sub, p_obj=p_obj, p_img=p_img, p_sub=p_sub, K=synth.K_MALAGA)


def _w_sharded_batch(rank, world):
    import oracle as o
    from visual_odometry_amd import parallel as par
    pr = _batch_problems()
    t = torch.from_numpy
    be = _OracleBackend(o)
    out = []
    for kind, sub, pts, img, thr in ((0, pr["f_sub"], pr["f_pts"], None, 1.0), (2, pr["p_sub"], pr["p_obj"], pr["p_img"], 2.0)):
        args = (kind, t(sub), t(pts), thr, 0.999, pr["H"])
        kw = dict(img=None if img is None else t(img), K=pr["K"], n_dev=t(pr["nd"]), backend=be)
        sharded = par.ransac_batch_sharded(*args, group=par.WORLD, **kw)
        local = par.ransac_batch_sharded(*args, group=None, **kw)          # group=None: no collective, all iterations here
        out.append(all(torch.equal(a, b) for a, b in zip(sharded, local) if a is not None))
        out.append([x.numpy().tolist() for x in sharded if x is not None])
    return out


def _w_gather_dim1(rank, world):
    from visual_odometry_amd import parallel as par
    ok = True
    for n in (12, 13):                                     # divisible and ragged
        full = torch.arange(2 * n * 4, dtype=torch.int32).reshape(2, n, 4)
        lo, hi = par.shard_bounds(n, rank, world)
        ok &= bool(torch.equal(par.all_gather_dim1(full[:, lo:hi].contiguous(), n, par.WORLD), full))
    x = torch.zeros((3, 2), dtype=torch.float64)
    if rank == world - 1:
        x[1, 0], x[2, 1] = -0.0, float("nan")
        x[0, 0] = 1.2345e-300
    y = par.all_reduce_sum_bits(x.clone(), par.WORLD)
    exp = torch.zeros((3, 2), dtype=torch.float64); exp[1, 0], exp[2, 1], exp[0, 0] = -0.0, float("nan"), 1.2345e-300
    ok &= bool(torch.equal(y.view(torch.int64), exp.view(torch.int64)))
    return ok


def _w_group_none_is_local(rank, world):
    from visual_odometry_amd import parallel as par
    return par.world_info(None) == (0, 1) and par.world_info(par.WORLD) == (rank, world) \
        and par.round_robin(list(range(10)), rank, world) == list(range(10))[rank::world]


# ------------------------------------------------------------------ tests
@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_rows(world):
    assert all(_run("_w_gather", world))


def test_allreduce_best_lowest_index_wins():
    assert all(_run("_w_best", 2))


def test_group_none_does_not_shard():
    assert all(_run("_w_group_none_is_local", 2))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ransac_equals_unsharded(oracle_mod, world):
    res = _run("_w_sharded_fund", world)
    p1, p2, thr, conf, iters = _fund_problem()
    F, mask, info = oracle_mod.find_fundamental_mat(p1, p2, thr, conf, iters)
    for r in res:
        assert r[:4] == tuple(int(v) for v in info)
        assert np.array_equal(np.array(r[4]).reshape(3, 3), F)


@pytest.mark.parametrize("world", [2, 3])
def test_gather_dim1_and_bit_sum(world):
    assert all(_run("_w_gather_dim1", world))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ransac_batch_equals_unsharded(oracle_mod, world):
    """parallel.ransac_batch_sharded (the batched, fused form bench.py --mode shard and pipeline.match_and_ransac_sharded
    use on the GPUs) with the oracle as per-rank backend: every rank must get the unsharded result, and that
    result must be the oracle's sequential RANSAC loop."""
    res = _run("_w_sharded_batch", world)
    pr = _batch_problems()
    for r in res:
        assert r[0] is True and r[2] is True
        assert r[1] == res[0][1] and r[3] == res[0][3]                 # identical on every rank
    result, best, mask = (np.array(x) for x in res[0][1])
    for b in range(pr["B"]):
        n = int(pr["nd"][b])
        F, omask, info = oracle_mod.find_fundamental_mat(pr["f_pts"][b, :n, :2], pr["f_pts"][b, :n, 2:], 1.0, 0.999, pr["H"])
        assert tuple(result[b]) == tuple(int(v) for v in info)
        assert np.array_equal(best[b].reshape(3, 3), F)
        assert np.array_equal(mask[b, :n], omask.ravel()) and not mask[b, n:].any()


@pytest.mark.parametrize("world", [2, 3])
def test_sweep_tables_gather(world):
    assert all(_run("_w_sweep_tables", world))
