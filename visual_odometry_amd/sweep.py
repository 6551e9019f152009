"""Offline sweep over many frame pairs (the caller side of cfg 5): synthetic frame
pairs are generated ON the device batch by batch, go through the device-resident pipeline
(match -> compaction -> getSubset -> F / PnP RANSAC -> 2D-2D pose tail) and leave one small row per
pair in a result table.  Pairs are dealt round-robin over the ranks (one process per GPU, no
data-path collective); the tables meet on rank 0 with one host-side gather.

The generators follow the recipes of SURVEY.md 8d (SIFT-like descriptors: Gamma(0.6) magnitudes,
normalise, clip 0.2, renormalise, x512, round, saturate; scene: points in the frustum at 4-40 m,
rotation <= 5 deg, unit baseline, pixel noise) with torch's device generator, seeded per pair, so a
pair's data does not depend on which rank or batch it lands in."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import parallel, pipeline, synth

TABLE_DTYPE = np.dtype([
    ("pair", np.int64), ("n_match", np.int32),
    ("f_iter", np.int32), ("f_model", np.int32), ("f_inliers", np.int32), ("f_niters", np.int32),
    ("p_iter", np.int32), ("p_model", np.int32), ("p_inliers", np.int32), ("p_niters", np.int32),
    ("tail_score", np.int32), ("tail_kept", np.int32),
    ("rot_err_deg", np.float64), ("t_dir_err_deg", np.float64),
])


def _gen(dev: torch.device, seed: int) -> torch.Generator:
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    return g


def sift_like_device(n: int, dev: torch.device) -> torch.Tensor:
    """(n,128) f32 integer-valued rows, the recipe of synth.sift_like.  torch's exact Gamma sampler
    draws from the device's default generator (seeded per pair by the caller)."""
    d = torch._standard_gamma(torch.full((n, 128), 0.6, device=dev, dtype=torch.float64))
    d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    d = d.clamp_max(0.2)
    d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return torch.clamp(torch.round(d * 512.0), 0, 255).to(torch.float32)


def _rodrigues(av: torch.Tensor) -> torch.Tensor:
    th = av.norm()
    k = av / th.clamp_min(1e-300)
    Kx = torch.zeros((3, 3), dtype=av.dtype, device=av.device)
    Kx[0, 1], Kx[0, 2], Kx[1, 0], Kx[1, 2], Kx[2, 0], Kx[2, 1] = -k[2], k[1], k[2], -k[0], -k[1], k[0]
    return torch.eye(3, dtype=av.dtype, device=av.device) + torch.sin(th) * Kx + (1 - torch.cos(th)) * (Kx @ Kx)


def frame_pair_device(pair_id: int, nq: int, nt: int, K: np.ndarray, size, dev: torch.device,
                      match_frac: float = 0.6, desc_noise: float = 8.0, noise_px: float = 0.3,
                      max_rot_deg: float = 5.0, seed0: int = 1234) -> Dict[str, torch.Tensor]:
    """One synthetic frame pair on the device (same structure as synth.frame_pair)."""
    g = _gen(dev, seed0 + pair_id)
    torch.cuda.manual_seed(seed0 + pair_id)                 # the Gamma sampler uses the device's default generator
    f64 = dict(generator=g, device=dev, dtype=torch.float64)
    ax = torch.randn(3, **f64)
    ax = ax / ax.norm()
    R = _rodrigues(ax * np.deg2rad(max_rot_deg) * (0.3 + 0.7 * torch.rand((), **f64)))
    t = torch.randn(3, **f64)
    t = t / t.norm()
    w, h = size
    Kt = torch.from_numpy(np.asarray(K, dtype=np.float64)).to(dev)
    z = 4.0 + 36.0 * torch.rand(nq, **f64)
    u = w * torch.rand(nq, **f64)
    v = h * torch.rand(nq, **f64)
    X = torch.stack([(u - Kt[0, 2]) / Kt[0, 0] * z, (v - Kt[1, 2]) / Kt[1, 1] * z, z], dim=1)
    Xc = X @ R.T + t
    p2 = Xc[:, :2] / Xc[:, 2:3] * torch.stack([Kt[0, 0], Kt[1, 1]]) + torch.stack([Kt[0, 2], Kt[1, 2]])
    p1 = torch.stack([u, v], dim=1)
    p1 = p1 + noise_px * torch.randn(p1.shape, **f64)
    p2 = p2 + noise_px * torch.randn(p2.shape, **f64)
    q = sift_like_device(nq, dev)
    tr = sift_like_device(nt, dev)
    nmatch = int(match_frac * min(nq, nt))
    src = torch.randperm(nq, generator=g, device=dev)[:nmatch]
    dst = torch.randperm(nt, generator=g, device=dev)[:nmatch]
    noisy = q[src].to(torch.float64) + desc_noise * torch.randn((nmatch, 128), **f64)
    tr[dst] = torch.clamp(torch.round(noisy), 0, 255).to(torch.float32)
    kp_t = torch.stack([w * torch.rand(nt, **f64), h * torch.rand(nt, **f64)], dim=1).to(torch.float32)
    kp_t[dst] = p2[src].to(torch.float32)
    return dict(q=q, t=tr, kp_q=p1.to(torch.float32), kp_t=kp_t, xyz_q=X.to(torch.float32), R=R, t_vec=t)


def _angle_deg(c: float) -> float:
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def run_sweep(pair_ids: Sequence[int], nq: int, nt: int, batch: int, dev: torch.device, K: np.ndarray = synth.K_MALAGA,
              size=synth.IMAGE_SIZE["malaga"], ratio: float = 0.8, f_threshold: float = 0.1, f_confidence: float = 0.9999,
              f_iters: int = 4096, p_threshold: float = 1.0, p_confidence: float = 0.9999, p_iters: int = 4096,
              first_block_pnp: int = 64, max_depth: float = 50.0, seed0: int = 1234, group=None) -> Optional[np.ndarray]:
    """Process this rank's share of `pair_ids` (round-robin over the group) in batches of `batch`
    pairs; returns the merged table (TABLE_DTYPE, sorted by pair id) on rank 0, None elsewhere.
    One host synchronisation per batch (to read the small result tensors)."""
    rank, world = parallel.world_info(group)
    mine = parallel.round_robin(list(pair_ids), rank, world)
    rows: List[np.ndarray] = []
    ws = None
    for b0 in range(0, len(mine), batch):
        ids = mine[b0:b0 + batch]
        pairs = [frame_pair_device(i, nq, nt, K, size, dev, seed0=seed0) for i in ids]
        st = lambda k: torch.stack([p[k] for p in pairs])
        if ws is None:
            ws = torch.empty((pipeline.workspace_bytes(nq, nt, batch, f_iters, p_iters),), dtype=torch.uint8, device=dev)
        r = pipeline.match_and_ransac(st("q"), st("t"), st("kp_q"), st("kp_t"), ratio, f_threshold, f_confidence, f_iters,
                                      xyz_q=st("xyz_q"), K=K, p_threshold=p_threshold, p_confidence=p_confidence,
                                      p_iters=p_iters, first_block_pnp=first_block_pnp,
                                      workspace=ws if len(ids) == batch else None)
        pipeline.finish_small_pairs(r, f_threshold, f_confidence, f_iters)      # host check point: raises on flagged descriptors
        tail = pipeline.pose_tail(r, K, max_depth)
        n = r.n_match.cpu().numpy()
        fr, pr = r.f_result.cpu().numpy(), r.p_result.cpu().numpy()
        sc, kept, T = tail["score"].cpu().numpy(), tail["keep"].sum(dim=1).cpu().numpy(), tail["T"].cpu().numpy()
        tab = np.zeros(len(ids), dtype=TABLE_DTYPE)
        tab["pair"], tab["n_match"] = ids, n
        for j, key in enumerate(("iter", "model", "inliers", "niters")):
            tab["f_" + key], tab["p_" + key] = fr[:, j], pr[:, j]
        tab["tail_score"], tab["tail_kept"] = sc.max(axis=1), kept
        for i, p in enumerate(pairs):
            Rg, tg = p["R"].cpu().numpy(), p["t_vec"].cpu().numpy()
            tab["rot_err_deg"][i] = _angle_deg((np.trace(T[i][:3, :3] @ Rg.T) - 1) / 2)
            tn = T[i][:3, 3] / max(np.linalg.norm(T[i][:3, 3]), 1e-300)
            tab["t_dir_err_deg"][i] = _angle_deg(float(tn @ tg))
        rows.append(tab)
    local = np.concatenate(rows) if rows else np.zeros(0, dtype=TABLE_DTYPE)
    return gather_tables(local, group)


def gather_tables(local: np.ndarray, group=None) -> Optional[np.ndarray]:
    """Host-side gather of the per-rank tables on rank 0 (sorted by pair id); no device collective."""
    rank, world = parallel.world_info(group)
    if world == 1:
        return np.sort(local, order="pair")
    import torch.distributed as dist
    g = None if group is parallel.WORLD else group
    parts: List[Optional[np.ndarray]] = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0, group=g)
    if rank != 0:
        return None
    return np.sort(np.concatenate(parts), order="pair")
