"""Multi-GPU layout of the hot path on one 8 x MI355X node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).

Nothing on this path needs a bulk exchange:
  * many frame pairs        -> pairs are dealt round-robin over ranks, no collective;
  * one big match           -> query rows are split over ranks, the train set is replicated (1 MB at
                               8192 rows); rows are independent, an all_gather only assembles the result;
  * one RANSAC problem      -> the hypothesis (sample) range is split over ranks, correspondences are
                               replicated (<= 640 KB).  Each rank solves and scores its slice, then ONE
                               all_gather of the int32 inlier counts (16-256 KB in total) gives every
                               rank the full count vector and the replay of OpenCV's sequential
                               best-model bookkeeping runs replicated -> exactly the single-GPU
                               answer.  The winning model (72-96 B) is broadcast from its owner.
    In pure-argmax mode a single packed int64 MAX all-reduce replaces the gather.
Messages are tiny, i.e. latency-bound: batch several problems into one collective where possible.
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
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=_pg(group))
    parts = []
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        parts.append(out[r * width: r * width + (b - a)])
    return torch.cat(parts, dim=0)


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
