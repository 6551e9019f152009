"""Multi-GPU layout of the hot path on one 8 x MI355X node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).

Nothing on this path needs a bulk exchange:
  * many frame pairs        -> pairs are dealt round-robin over ranks, no collective;
  * one big match           -> query rows are split over ranks, the train set is replicated (1 MB at
                               8192 rows); rows are independent, an all_gather only assembles the result;
  * one RANSAC problem      -> the hypothesis (sample) range is split over ranks, correspondences are
                               replicated (<= 640 KB).  Each rank solves and scores its slice, then ONE
                               all_gather of the int32 inlier counts (per PROBLEM 16-256 KB in total; the
                               batched form gathers B x H x (width + 1) int32 = 16.8 MB per cfg4 step) gives
                               every rank the full count vector and the replay of OpenCV's sequential
                               best-model bookkeeping runs replicated -> exactly the single-GPU
                               answer.  The winning model (72-96 B) is broadcast from its owner.
    In pure-argmax mode a single packed int64 MAX all-reduce replaces the gather.
Per problem the messages are tiny; batched over a step's 256 problems they reach 4-17 MB per collective
(sizes in DESIGN.md section 6) -- whether that is latency- or link-bound on xGMI is unmeasured.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


WORLD = "world"      # pass as `group` to shard over every rank of the default process group


def _pg(group):
    return None if group == WORLD else group


def world_info(group=None) -> Tuple[int, int]:
    """(rank, world) of `group`.  group=None means "do not shard" -> (0, 1), even inside a
    distributed job: ranks that work on different frame pairs must not meet in a collective."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(_pg(group)), dist.get_world_size(_pg(group))


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of range(n) owned by `rank`; sizes differ by at most one and
    earlier ranks get the larger slices (so the owner of index i is cheap to compute)."""
    base, rem = divmod(max(n, 0), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(i: int, n: int, world: int) -> int:
    base, rem = divmod(max(n, 0), world)
    edge = rem * (base + 1)
    return i // (base + 1) if i < edge else rem + (i - edge) // max(base, 1)


def round_robin(items: Sequence, rank: int, world: int) -> List:
    """Frame pairs (or any independent work items) of this rank."""
    return list(items[rank::world])


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate the ranks' row slices (made with shard_bounds over n_total rows) along dim 0.
    Slices are padded to the largest slice so one fixed-size all_gather suffices."""
    rank, world = world_info(group)
    if world == 1:
        return local
    width = -(-n_total // world)
    lo, hi = shard_bounds(n_total, rank, world)
    assert local.shape[0] == hi - lo, "local slice does not match shard_bounds"
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    out = all_gather_stack(pad, group).reshape((world * width,) + tuple(local.shape[1:]))
    if n_total == world * width:
        return out
    return out.index_select(0, _gather_index(n_total, world, out.device))


def pack_best(count: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """(count, index) -> int64 key whose maximum is the largest count at the LOWEST index."""
    return (count.to(torch.int64) << 32) | (0xFFFFFFFF - index.to(torch.int64))


def unpack_best(key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return (key >> 32).to(torch.int32), (0xFFFFFFFF - (key & 0xFFFFFFFF)).to(torch.int32)


def allreduce_best(local_count: torch.Tensor, local_global_index: torch.Tensor, group=None):
    """Pure-argmax reduction: every rank contributes its best (count, global hypothesis index); all
    ranks receive the global best with ties going to the lowest index.  One 8-byte MAX all-reduce
    (per problem: pass 1-D tensors to reduce many problems in one call)."""
    key = pack_best(local_count, local_global_index)
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(key, op=dist.ReduceOp.MAX, group=_pg(group))
    return unpack_best(key)


def broadcast_from_owner(value: torch.Tensor, owner: int, group=None) -> torch.Tensor:
    _, world = world_info(group)
    if world > 1:
        pg = _pg(group)
        dist.broadcast(value, src=dist.get_global_rank(pg, owner) if pg is not None else owner, group=pg)
    return value


def sharded_ransac(n_iters: int, max_models: int, model_width: int,
                   solve_and_score: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                   replay: Callable[[torch.Tensor, torch.Tensor], Tuple[int, int, int, int]],
                   group=None):
    """Hypothesis-sharded RANSAC skeleton.

    solve_and_score(lo, hi) -> (counts (hi-lo, max_models) i32, nmodels (hi-lo,) i32,
                                models (hi-lo, max_models, model_width) f64) for samples lo..hi-1;
    replay(counts (n_iters, max_models), nmodels (n_iters,)) -> (best_iter, best_model, max_good, niters).
    Returns (best_iter, best_model, max_good, niters, model (model_width,) f64 | None) on every rank."""
    rank, world = world_info(group)
    lo, hi = shard_bounds(n_iters, rank, world)
    counts, nmodels, models = solve_and_score(lo, hi)
    packed = torch.cat([counts.reshape(hi - lo, max_models), nmodels.reshape(hi - lo, 1)], dim=1)
    full = all_gather_rows(packed.contiguous(), n_iters, group)
    best_iter, best_k, max_good, niters = replay(full[:, :max_models].contiguous(), full[:, max_models].contiguous())
    if best_iter < 0:
        return best_iter, best_k, max_good, niters, None
    owner = owner_of(best_iter, n_iters, world)
    if rank == owner:
        model = models.reshape(hi - lo, max_models, model_width)[best_iter - lo, best_k].clone()
    else:
        model = torch.empty((model_width,), dtype=models.dtype, device=models.device)
    return best_iter, best_k, max_good, niters, broadcast_from_owner(model, owner, group)


# ------------------------------------------------------------------ batched, hypothesis-sharded RANSAC
def _is_gloo(group) -> bool:
    return dist.get_backend(_pg(group)) == "gloo"


def _live(group) -> bool:
    """True when `group` names a process group: the collectives below then run even with one rank (a one-rank
    RCCL job, e.g. `bench.py --gpus 1`, goes through the same calls as an eight-rank one)."""
    return group is not None and dist.is_available() and dist.is_initialized()


def all_gather_stack(local: torch.Tensor, group=None) -> torch.Tensor:
    """(world,) + local.shape: every rank's `local` (same shape and dtype everywhere), ONE collective.
    RCCL gathers device tensors in place; under gloo (CPU tests, or several ranks on one GPU) device
    tensors take the round trip through host memory."""
    _, world = world_info(group)
    if not _live(group):
        return local[None]
    local = local.contiguous()
    if local.is_cuda and _is_gloo(group):
        host = local.cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=_pg(group))
        return out.reshape((world,) + tuple(host.shape)).to(local.device)
    # output laid out as the concatenation along dim 0 (the form every backend accepts), viewed as a stack
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=_pg(group))
    return out.reshape((world,) + tuple(local.shape))


def all_reduce_sum_bits(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place integer SUM of the raw 8-byte words of `t` (f64 or i64).  With all-zero bytes on every rank
    but one, every rank ends up with that rank's exact bit pattern (also for -0.0 and NaN payloads)."""
    if not _live(group):
        return t
    bits = t.view(torch.int64)
    if bits.is_cuda and _is_gloo(group):
        host = bits.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=_pg(group))
        bits.copy_(host)
    else:
        dist.all_reduce(bits, op=dist.ReduceOp.SUM, group=_pg(group))
    return t


def all_gather_dim1(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local (B, n_local, ...) = this rank's shard_bounds(n_total) slice along dim 1 -> (B, n_total, ...)."""
    rank, world = world_info(group)
    if not _live(group):
        return local
    width = -(-n_total // world)
    lo, hi = shard_bounds(n_total, rank, world)
    assert local.shape[1] == hi - lo, "local slice does not match shard_bounds"
    if hi - lo == width:
        pad = local
    else:
        pad = torch.zeros((local.shape[0], width) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)
        pad[:, : hi - lo] = local
    g = all_gather_stack(pad, group)                        # (world, B, width, ...)
    if n_total == world * width:
        return g.transpose(0, 1).reshape((local.shape[0], n_total) + tuple(local.shape[2:]))
    flat = g.transpose(0, 1).reshape((local.shape[0], world * width) + tuple(local.shape[2:]))
    return flat.index_select(1, _gather_index(n_total, world, flat.device))


_gather_idx = {}


def _gather_index(n: int, world: int, device) -> torch.Tensor:
    """Row of the rank-major padded gather buffer ((world * ceil(n / world)) rows) that holds global row i, for
    every i < n: rank r's slice shard_bounds(n, r, world) starts at r * ceil(n / world)."""
    key = (n, world, str(device))
    if key not in _gather_idx:
        per = -(-n // world)
        idx = torch.empty((n,), dtype=torch.int64)
        for r in range(world):
            a, b = shard_bounds(n, r, world)
            idx[a:b] = torch.arange(r * per, r * per + (b - a))
        _gather_idx[key] = idx.to(device)
    return _gather_idx[key]


class DeviceRansacBackend:
    """The three per-rank steps of the sharded batch on the MI355X (visual_odometry_amd.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def slice_counts(self, kind, samples, pts, threshold, lo, hi, img, K, n_dev, workspace):
        ws = self.ops.ransac_slice_counts(kind, samples, pts, threshold, lo, hi, img=img, K=K, n_dev=n_dev, workspace=workspace)
        counts, nmodels = self.ops.ransac_tables(kind, samples.shape[1], samples.shape[0], ws)
        return counts, nmodels, ws

    def slice_finish(self, kind, H, lo, hi, N, batch, confidence, max_iters, state, n_dev):
        return self.ops.ransac_slice_finish(kind, H, lo, hi, N, batch, confidence, max_iters, state, n_dev=n_dev)

    def mask_models(self, kind, best, result, pts, threshold, img, K, n_dev):
        return self.ops.ransac_mask_models(kind, best, result, pts, threshold, img=img, K=K, n_dev=n_dev)


def ransac_batch_sharded(kind: int, samples: torch.Tensor, pts: torch.Tensor, threshold: float, confidence: float,
                         max_iters: int, img: Optional[torch.Tensor] = None, K=None, n_dev: Optional[torch.Tensor] = None,
                         workspace: Optional[torch.Tensor] = None, group=WORLD, backend=None):
    """ops.ransac_batch (first_block = 0) with the iteration range of EVERY problem of the batch split over the
    ranks of `group`; all ranks pass identical inputs and receive identical outputs, bit for bit those of the
    unsharded call:  result (B,4) i32, best_model (B,9|12) f64, best_rvec (B,3) f64 | None, mask (B,N) u8.

    Per rank: solve + score its iterations (backend.slice_counts) -> ONE all_gather of the batch's count-table
    slices (counts and nmodels packed in one int32 tensor) -> replicated replay of OpenCV's sequential
    bookkeeping (backend.slice_finish) -> ONE integer-SUM all-reduce that hands every rank the winners' model
    bits -> masks (backend.mask_models, replicated).  `backend` is DeviceRansacBackend unless a test
    substitutes a CPU one."""
    be = backend or DeviceRansacBackend()
    rank, world = world_info(group)
    B, H, N = samples.shape[0], samples.shape[1], pts.shape[1]
    lo, hi = shard_bounds(H, rank, world)
    counts, nmodels, state = be.slice_counts(kind, samples, pts, threshold, lo, hi, img, K, n_dev, workspace)
    if _live(group):
        width = counts.shape[2]
        per = -(-H // world)
        send = torch.zeros((B, per, width + 1), dtype=torch.int32, device=counts.device)
        send[:, : hi - lo, :width] = counts[:, lo:hi]
        send[:, : hi - lo, width] = nmodels[:, lo:hi]
        got = all_gather_stack(send, group)                 # (world, B, per, width + 1)
        # one gather along the iteration axis puts every rank's rows where the tables want them (the padded
        # tail rows of the shorter slices are never indexed); the tables live in the workspace, so copy in place
        full = got.permute(1, 0, 2, 3).reshape(B, world * per, width + 1).index_select(1, _gather_index(H, world, counts.device))
        counts.copy_(full[:, :, :width])
        nmodels.copy_(full[:, :, width])
    result, best, rvec = be.slice_finish(kind, H, lo, hi, N, B, confidence, max_iters, state, n_dev)
    if _live(group):
        if rvec is not None:
            both = torch.cat([best, rvec], dim=1).contiguous()
            all_reduce_sum_bits(both, group)
            best, rvec = both[:, : best.shape[1]].contiguous(), both[:, best.shape[1]:].contiguous()
        else:
            all_reduce_sum_bits(best, group)
    mask = be.mask_models(kind, best, result, pts, threshold, img, K, n_dev)
    return result, best, rvec, mask
