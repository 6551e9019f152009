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
def test_sweep_tables_gather(world):
    assert all(_run("_w_sweep_tables", world))
