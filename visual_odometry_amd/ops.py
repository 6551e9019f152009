"""Device-level operators: torch tensors on an MI355X in, torch tensors out, no host sync.

Thin, typed wrappers over the C ABI of ``libvo_hip.so`` (``include/vo_hip.h``).  torch is used for
device memory and the current HIP stream only; all arithmetic happens in the library's kernels.
Every function launches on ``torch.cuda.current_stream()`` and returns immediately.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

_INT_MAX = 2**31 - 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.VoError(f"{name} must live on the GPU")
    if t.dtype != dtype:
        raise _lib.VoError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise _lib.VoError(f"{name} lives on cuda:{t.device.index} but the current device is "
                           f"cuda:{torch.cuda.current_device()}: launches go to the current device's stream")
    return t.contiguous()


def _K4(K) -> "C.Array":
    K = np.asarray(K, dtype=np.float64)
    if K.shape == (3, 3):
        vals = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    else:
        vals = tuple(K.reshape(4))
    return (C.c_double * 4)(*vals)


def thr2_of(threshold: float) -> float:
    """findInliers' ``(float)(threshold*threshold)`` (f64 product rounded once to f32)."""
    return float(np.float32(float(threshold) * float(threshold)))


# ------------------------------------------------------------------------------- matching
def desc_to_i8(desc: torch.Tensor, flags: Optional[torch.Tensor] = None,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(n,128) f32 integer-valued descriptors -> (int8 rows, int32 squared norms, int32 flag word).
    `flags` (1,) int32, already zero, may be passed in (the kernel only ORs into it); `out` = ((n,128) i8, (n,) i32)
    buffers to write into instead of new ones."""
    _lib.require_device()
    desc = _dev(desc, torch.float32, "desc")
    if desc.ndim != 2 or desc.shape[1] != 128:
        raise _lib.VoError(f"descriptors must be (n,128), got {tuple(desc.shape)}")
    n = desc.shape[0]
    if out is not None:
        out, norm = _dev(out[0], torch.int8, "out rows"), _dev(out[1], torch.int32, "out norms")
        if tuple(out.shape) != (n, 128) or tuple(norm.shape) != (n,):
            raise _lib.VoError("desc_to_i8: out buffers must be (n,128) int8 and (n,) int32")
    else:
        out = torch.empty((n, 128), dtype=torch.int8, device=desc.device)
        norm = torch.empty((n,), dtype=torch.int32, device=desc.device)
    if flags is None:
        flags = torch.zeros((1,), dtype=torch.int32, device=desc.device)
    else:
        flags = _dev(flags, torch.int32, "flags")
    _lib.check(_lib.load().vo_desc_f32_to_i8(_ptr(desc), n, _ptr(out), _ptr(norm), _ptr(flags), _stream()),
               "vo_desc_f32_to_i8")
    return out, norm, flags


def knn2(q: torch.Tensor, qnorm: torch.Tensor, t: torch.Tensor, tnorm: torch.Tensor,
         workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """q (B,nq,128)|(nq,128) int8, t likewise -> idx (...,nq,2) i32, d2 (...,nq,2) i32 (exact)."""
    _lib.require_device()
    lib = _lib.load()
    q = _dev(q, torch.int8, "q"); t = _dev(t, torch.int8, "t")
    qnorm = _dev(qnorm, torch.int32, "qnorm"); tnorm = _dev(tnorm, torch.int32, "tnorm")
    batched = q.ndim == 3
    q3 = q if batched else q[None]
    t3 = t if batched else t[None]
    B, nq, nt = q3.shape[0], q3.shape[1], t3.shape[1]
    if t3.shape[0] != B or q3.shape[2] != 128 or t3.shape[2] != 128:
        raise _lib.VoError("knn2: shape mismatch")
    idx = torch.empty((B, nq, 2), dtype=torch.int32, device=q.device)
    d2 = torch.empty((B, nq, 2), dtype=torch.int32, device=q.device)
    need = lib.vo_knn2_workspace_bytes(nq, nt, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 15) // 16 * 16,), dtype=torch.uint8, device=q.device)
    _lib.check(lib.vo_knn2_l2_i8(_ptr(q3), _ptr(qnorm), nq, _ptr(t3), _ptr(tnorm), nt, _ptr(idx), _ptr(d2), B,
                                 _ptr(workspace), workspace.numel() * workspace.element_size(), _stream()),
               "vo_knn2_l2_i8")
    return (idx, d2) if batched else (idx[0], d2[0])


def ratio_test(d2: torch.Tensor, ratio: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """d2 (n,2) i32 -> keep (n,) u8, count (1,) i32."""
    _lib.require_device()
    d2 = _dev(d2, torch.int32, "d2")
    n = d2.shape[0]
    keep = torch.empty((n,), dtype=torch.uint8, device=d2.device)
    count = torch.zeros((1,), dtype=torch.int32, device=d2.device)
    _lib.check(_lib.load().vo_ratio_test(_ptr(d2), n, float(ratio), _ptr(keep), _ptr(count), _stream()),
               "vo_ratio_test")
    return keep, count


# ------------------------------------------------------------------------------- scoring
def _valid(valid, H):
    if valid is None:
        return None
    v = _dev(valid, torch.uint8, "valid")
    if v.numel() != H:
        raise _lib.VoError("valid must have one byte per model")
    return v


def score_fund(models: torch.Tensor, pts4: torch.Tensor, threshold: float,
               valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    _lib.require_device()
    models = _dev(models, torch.float64, "models").reshape(-1, 9)
    pts4 = _dev(pts4, torch.float32, "pts4").reshape(-1, 4)
    H, N = models.shape[0], pts4.shape[0]
    counts = torch.empty((H,), dtype=torch.int32, device=models.device)
    _lib.check(_lib.load().vo_score_fund(_ptr(models), _ptr(_valid(valid, H)), H, _ptr(pts4), N,
                                         thr2_of(threshold), _ptr(counts), _stream()), "vo_score_fund")
    return counts


def score_fund_bound(models: torch.Tensor, pts4: torch.Tensor, threshold: float) -> torch.Tensor:
    """Upper bounds of score_fund's counts from the F count kernel's first (matrix-core) tier alone: the correspondences it
    cannot prove to be outliers (vo_score_fund_bound; what vo_ransac_batch prunes hopeless hypotheses with)."""
    _lib.require_device()
    models = _dev(models, torch.float64, "models").reshape(-1, 9)
    pts4 = _dev(pts4, torch.float32, "pts4").reshape(-1, 4)
    H, N = models.shape[0], pts4.shape[0]
    bound = torch.empty((H,), dtype=torch.int32, device=models.device)
    _lib.check(_lib.load().vo_score_fund_bound(_ptr(models), H, _ptr(pts4), N, thr2_of(threshold), _ptr(bound), _stream()),
               "vo_score_fund_bound")
    return bound


def score_ess(models: torch.Tensor, pts4: torch.Tensor, threshold: float,
              valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    _lib.require_device()
    models = _dev(models, torch.float64, "models").reshape(-1, 9)
    pts4 = _dev(pts4, torch.float64, "pts4").reshape(-1, 4)
    H, N = models.shape[0], pts4.shape[0]
    counts = torch.empty((H,), dtype=torch.int32, device=models.device)
    _lib.check(_lib.load().vo_score_ess(_ptr(models), _ptr(_valid(valid, H)), H, _ptr(pts4), N,
                                        thr2_of(threshold), _ptr(counts), _stream()), "vo_score_ess")
    return counts


def score_pnp(models: torch.Tensor, obj: torch.Tensor, img: torch.Tensor, K, threshold: float,
              valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    _lib.require_device()
    models = _dev(models, torch.float64, "models").reshape(-1, 12)
    obj = _dev(obj, torch.float32, "obj").reshape(-1, 3)
    img = _dev(img, torch.float32, "img").reshape(-1, 2)
    H, N = models.shape[0], obj.shape[0]
    if img.shape[0] != N:
        raise _lib.VoError("score_pnp: obj/img length mismatch")
    counts = torch.empty((H,), dtype=torch.int32, device=models.device)
    _lib.check(_lib.load().vo_score_pnp(_ptr(models), _ptr(_valid(valid, H)), H, _ptr(obj), _ptr(img), _K4(K), N,
                                        thr2_of(threshold), _ptr(counts), _stream()), "vo_score_pnp")
    return counts


def mask_fund(model: torch.Tensor, pts4: torch.Tensor, threshold: float) -> torch.Tensor:
    _lib.require_device()
    model = _dev(model, torch.float64, "model").reshape(9)
    pts4 = _dev(pts4, torch.float32, "pts4").reshape(-1, 4)
    mask = torch.empty((pts4.shape[0],), dtype=torch.uint8, device=pts4.device)
    _lib.check(_lib.load().vo_mask_fund(_ptr(model), _ptr(pts4), pts4.shape[0], thr2_of(threshold), _ptr(mask),
                                        _stream()), "vo_mask_fund")
    return mask


def mask_ess(model: torch.Tensor, pts4: torch.Tensor, threshold: float) -> torch.Tensor:
    _lib.require_device()
    model = _dev(model, torch.float64, "model").reshape(9)
    pts4 = _dev(pts4, torch.float64, "pts4").reshape(-1, 4)
    mask = torch.empty((pts4.shape[0],), dtype=torch.uint8, device=pts4.device)
    _lib.check(_lib.load().vo_mask_ess(_ptr(model), _ptr(pts4), pts4.shape[0], thr2_of(threshold), _ptr(mask),
                                       _stream()), "vo_mask_ess")
    return mask


def mask_pnp(model: torch.Tensor, obj: torch.Tensor, img: torch.Tensor, K, threshold: float) -> torch.Tensor:
    _lib.require_device()
    model = _dev(model, torch.float64, "model").reshape(12)
    obj = _dev(obj, torch.float32, "obj").reshape(-1, 3)
    img = _dev(img, torch.float32, "img").reshape(-1, 2)
    mask = torch.empty((obj.shape[0],), dtype=torch.uint8, device=obj.device)
    _lib.check(_lib.load().vo_mask_pnp(_ptr(model), _ptr(obj), _ptr(img), _K4(K), obj.shape[0],
                                       thr2_of(threshold), _ptr(mask), _stream()), "vo_mask_pnp")
    return mask


# ------------------------------------------------------------------------------- solvers
def solve_7pt(samples: torch.Tensor, pts4: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """samples (H,7) i32, pts4 (N,4) f32 -> models (H,3,9) f64, nmodels (H,) i32."""
    _lib.require_device()
    samples = _dev(samples, torch.int32, "samples").reshape(-1, 7)
    pts4 = _dev(pts4, torch.float32, "pts4").reshape(-1, 4)
    H = samples.shape[0]
    models = torch.zeros((H, 3, 9), dtype=torch.float64, device=pts4.device)
    nm = torch.zeros((H,), dtype=torch.int32, device=pts4.device)
    _lib.check(_lib.load().vo_solve_7pt(_ptr(samples), H, _ptr(pts4), pts4.shape[0], _ptr(models), _ptr(nm),
                                        _stream()), "vo_solve_7pt")
    return models, nm


def solve_5pt(samples: torch.Tensor, pts4: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """samples (H,5) i32, pts4 (N,4) f64 normalised -> models (H,10,9) f64, nmodels (H,) i32."""
    _lib.require_device()
    samples = _dev(samples, torch.int32, "samples").reshape(-1, 5)
    pts4 = _dev(pts4, torch.float64, "pts4").reshape(-1, 4)
    H = samples.shape[0]
    models = torch.zeros((H, 10, 9), dtype=torch.float64, device=pts4.device)
    nm = torch.zeros((H,), dtype=torch.int32, device=pts4.device)
    _lib.check(_lib.load().vo_solve_5pt(_ptr(samples), H, _ptr(pts4), pts4.shape[0], _ptr(models), _ptr(nm),
                                        _stream()), "vo_solve_5pt")
    return models, nm


def solve_p3p(samples: torch.Tensor, obj: torch.Tensor, img: torch.Tensor, K):
    """samples (H,4) i32 -> models (H,12) f64 [R|t], rvecs (H,3) f64, nmodels (H,) i32."""
    _lib.require_device()
    samples = _dev(samples, torch.int32, "samples").reshape(-1, 4)
    obj = _dev(obj, torch.float32, "obj").reshape(-1, 3)
    img = _dev(img, torch.float32, "img").reshape(-1, 2)
    H = samples.shape[0]
    models = torch.zeros((H, 12), dtype=torch.float64, device=obj.device)
    rvecs = torch.zeros((H, 3), dtype=torch.float64, device=obj.device)
    nm = torch.zeros((H,), dtype=torch.int32, device=obj.device)
    _lib.check(_lib.load().vo_solve_p3p(_ptr(samples), H, _ptr(obj), _ptr(img), _K4(K), obj.shape[0],
                                        _ptr(models), _ptr(rvecs), _ptr(nm), _stream()), "vo_solve_p3p")
    return models, rvecs, nm


# ------------------------------------------------------------------------------- bookkeeping
def ransac_replay(counts: torch.Tensor, nmodels: Optional[torch.Tensor], n_points: int, model_points: int,
                  confidence: float, max_iters: int) -> torch.Tensor:
    """counts (n_iters,max_models) i32 -> (4,) i32 [best_iter, best_model, max_good, niters]."""
    _lib.require_device()
    counts = _dev(counts, torch.int32, "counts")
    if counts.ndim == 1:
        counts = counts[:, None]
    n_iters, max_models = counts.shape
    nm = None if nmodels is None else _dev(nmodels, torch.int32, "nmodels")
    result = torch.empty((4,), dtype=torch.int32, device=counts.device)
    _lib.check(_lib.load().vo_ransac_replay(_ptr(counts), _ptr(nm), n_iters, max_models, n_points, model_points,
                                            float(confidence), int(max_iters), _ptr(result), _stream()),
               "vo_ransac_replay")
    return result


def argmax_first(counts: torch.Tensor) -> torch.Tensor:
    _lib.require_device()
    counts = _dev(counts, torch.int32, "counts").reshape(-1)
    result = torch.empty((2,), dtype=torch.int32, device=counts.device)
    _lib.check(_lib.load().vo_argmax_first(_ptr(counts), counts.numel(), _ptr(result), _stream()), "vo_argmax_first")
    return result


KIND_FUND, KIND_ESS, KIND_PNP = 0, 1, 2
_SAMPLE_WIDTH = {KIND_FUND: 7, KIND_ESS: 5, KIND_PNP: 4}


def ransac_batch(kind: int, samples: torch.Tensor, pts: torch.Tensor, threshold: float, confidence: float,
                 max_iters: int, img: Optional[torch.Tensor] = None, K=None, n_dev: Optional[torch.Tensor] = None,
                 workspace: Optional[torch.Tensor] = None, first_block: int = 0, static_scratch: bool = False):
    """Fused solve -> score -> replay -> mask for a batch of same-shape problems, no host sync.
    static_scratch=True goes through vo_ransac_batch_static: the library's built-in per-GPU scratch array
    instead of a workspace tensor (sizes beyond it raise; concurrent calls on one GPU race).

    samples (B,H,m) i32; pts (B,N,4) f32 [F] | (B,N,4) f64 [E] | (B,N,3) f32 object points [PnP, with
    img (B,N,2) f32 and K].  first_block > 0: iterations are evaluated in growing blocks and problems
    whose adaptive iteration count is already reached skip the rest (same results).  F: behind the first block only the
    hypotheses whose upper bound (vo_score_fund_bound) beats the best inlier count so far are counted exactly -- OpenCV's
    loop acts on goodCount > maxGoodCount only, so result record, model and mask are unchanged (1.5x faster at
    8192 x 8192 features with first_block = 256).
    -> result (B,4) i32, best_model (B,9|12) f64, best_rvec (B,3) f64 | None,
    mask (B,N) u8."""
    _lib.require_device()
    lib = _lib.load()
    m = _SAMPLE_WIDTH[kind]
    samples = _dev(samples, torch.int32, "samples")
    if samples.ndim != 3 or samples.shape[2] != m:
        raise _lib.VoError(f"samples must be (B,H,{m})")
    B, H = samples.shape[0], samples.shape[1]
    pts = _dev(pts, torch.float64 if kind == KIND_ESS else torch.float32, "pts")
    last = 3 if kind == KIND_PNP else 4
    if pts.ndim != 3 or pts.shape[0] != B or pts.shape[2] != last:
        raise _lib.VoError(f"pts must be (B,N,{last})")
    N = pts.shape[1]
    img_p = k4 = None
    if kind == KIND_PNP:
        img = _dev(img, torch.float32, "img")
        if tuple(img.shape) != (B, N, 2):
            raise _lib.VoError("img must be (B,N,2)")
        img_p, k4 = _ptr(img), _K4(K)
    nd = None if n_dev is None else _dev(n_dev, torch.int32, "n_dev")
    dev = samples.device
    result = torch.empty((B, 4), dtype=torch.int32, device=dev)
    mw = 12 if kind == KIND_PNP else 9
    best = torch.empty((B, mw), dtype=torch.float64, device=dev)
    rvec = torch.empty((B, 3), dtype=torch.float64, device=dev) if kind == KIND_PNP else None
    mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
    if static_scratch:
        _lib.check(lib.vo_ransac_batch_static(kind, _ptr(samples), H, _ptr(pts), img_p, k4, N, _ptr(nd), B, thr2_of(threshold),
                                              float(confidence), int(max_iters), int(first_block), _ptr(result), _ptr(best),
                                              _ptr(rvec), _ptr(mask), _stream()), "vo_ransac_batch_static")
        return result, best, rvec, mask
    need = lib.vo_ransac_workspace_bytes(kind, H, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=dev)
    _lib.check(lib.vo_ransac_batch(kind, _ptr(samples), H, _ptr(pts), img_p, k4, N, _ptr(nd), B, thr2_of(threshold),
                                   float(confidence), int(max_iters), int(first_block), _ptr(result), _ptr(best), _ptr(rvec), _ptr(mask),
                                   _ptr(workspace), workspace.numel() * workspace.element_size(), _stream()),
               "vo_ransac_batch")
    return result, best, rvec, mask


def _ransac_problem(kind, samples, pts, img, K, n_dev):
    """shape checks shared by the sliced entry points -> (samples, pts, img ptr, K4, n_dev, B, H, N)"""
    m = _SAMPLE_WIDTH[kind]
    samples = _dev(samples, torch.int32, "samples")
    if samples.ndim != 3 or samples.shape[2] != m:
        raise _lib.VoError(f"samples must be (B,H,{m})")
    B, H = samples.shape[0], samples.shape[1]
    pts = _dev(pts, torch.float64 if kind == KIND_ESS else torch.float32, "pts")
    last = 3 if kind == KIND_PNP else 4
    if pts.ndim != 3 or pts.shape[0] != B or pts.shape[2] != last:
        raise _lib.VoError(f"pts must be (B,N,{last})")
    N = pts.shape[1]
    img_t = k4 = None
    if kind == KIND_PNP:
        img_t = _dev(img, torch.float32, "img")
        if tuple(img_t.shape) != (B, N, 2):
            raise _lib.VoError("img must be (B,N,2)")
        k4 = _K4(K)
    nd = None if n_dev is None else _dev(n_dev, torch.int32, "n_dev")
    return samples, pts, img_t, k4, nd, B, H, N


def ransac_layout(kind: int, H: int, batch: int):
    """-> (byte offset of the nmodels table, of the counts table, models per sample, workspace bytes)"""
    off = (C.c_size_t * 4)()
    _lib.check(_lib.load().vo_ransac_layout(int(kind), int(H), int(batch), C.addressof(off)), "vo_ransac_layout")
    return int(off[0]), int(off[1]), int(off[2]), int(off[3])


def ransac_tables(kind: int, H: int, batch: int, workspace: torch.Tensor):
    """Views of the count tables inside a vo_ransac_* workspace: counts (B,H,width) i32, nmodels (B,H) i32."""
    o_nm, o_cnt, width, total = ransac_layout(kind, H, batch)
    flat = workspace.reshape(-1).view(torch.uint8)
    if flat.numel() < total:
        raise _lib.VoError("ransac_tables: workspace too small")
    nm = flat[o_nm:o_nm + 4 * batch * H].view(torch.int32).reshape(batch, H)
    cnt = flat[o_cnt:o_cnt + 4 * batch * H * width].view(torch.int32).reshape(batch, H, width)
    return cnt, nm


def ransac_slice_counts(kind: int, samples: torch.Tensor, pts: torch.Tensor, threshold: float, h_lo: int, h_hi: int,
                        img: Optional[torch.Tensor] = None, K=None, n_dev: Optional[torch.Tensor] = None,
                        workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Solve + score iterations [h_lo, h_hi) of every problem into `workspace` (allocated when None or too small;
    returned).  See vo_ransac_slice_counts in include/vo_hip.h."""
    _lib.require_device()
    lib = _lib.load()
    samples, pts, img_t, k4, nd, B, H, N = _ransac_problem(kind, samples, pts, img, K, n_dev)
    need = lib.vo_ransac_workspace_bytes(kind, H, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=samples.device)
    _lib.check(lib.vo_ransac_slice_counts(kind, _ptr(samples), H, int(h_lo), int(h_hi), _ptr(pts), _ptr(img_t), k4, N, _ptr(nd),
                                          B, thr2_of(threshold), _ptr(workspace),
                                          workspace.numel() * workspace.element_size(), _stream()), "vo_ransac_slice_counts")
    return workspace


def ransac_slice_finish(kind: int, H: int, h_lo: int, h_hi: int, N: int, batch: int, confidence: float, max_iters: int,
                        workspace: torch.Tensor, n_dev: Optional[torch.Tensor] = None):
    """Replay over the complete tables of `workspace` -> result (B,4) i32, best_model (B,9|12) f64 and best_rvec
    (B,3) f64 | None, the latter two zero unless this rank solved the winning iteration."""
    _lib.require_device()
    dev = workspace.device
    nd = None if n_dev is None else _dev(n_dev, torch.int32, "n_dev")
    result = torch.empty((batch, 4), dtype=torch.int32, device=dev)
    best = torch.empty((batch, 12 if kind == KIND_PNP else 9), dtype=torch.float64, device=dev)
    rvec = torch.empty((batch, 3), dtype=torch.float64, device=dev) if kind == KIND_PNP else None
    _lib.check(_lib.load().vo_ransac_slice_finish(kind, int(H), int(h_lo), int(h_hi), int(N), _ptr(nd), int(batch),
                                                  float(confidence), int(max_iters), _ptr(workspace),
                                                  workspace.numel() * workspace.element_size(), _ptr(result), _ptr(best),
                                                  _ptr(rvec), _stream()), "vo_ransac_slice_finish")
    return result, best, rvec


def ransac_mask_models(kind: int, best_model: torch.Tensor, result: torch.Tensor, pts: torch.Tensor, threshold: float,
                       img: Optional[torch.Tensor] = None, K=None, n_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mask (B,N) u8 of best_model[b] over problem b's correspondences (zeros where result[b,0] < 0)."""
    _lib.require_device()
    best_model = _dev(best_model, torch.float64, "best_model"); result = _dev(result, torch.int32, "result")
    pts = _dev(pts, torch.float64 if kind == KIND_ESS else torch.float32, "pts")
    B, N = pts.shape[0], pts.shape[1]
    img_p = k4 = None
    if kind == KIND_PNP:
        img_p, k4 = _ptr(_dev(img, torch.float32, "img")), _K4(K)
    nd = None if n_dev is None else _dev(n_dev, torch.int32, "n_dev")
    mask = torch.empty((B, N), dtype=torch.uint8, device=pts.device)
    _lib.check(_lib.load().vo_ransac_mask_models(kind, _ptr(best_model), _ptr(result), _ptr(pts), img_p, k4, N, _ptr(nd), B,
                                                 thr2_of(threshold), _ptr(mask), _stream()), "vo_ransac_mask_models")
    return mask


def fund_small_batch(pts4: torch.Tensor, n_match: torch.Tensor, confidence: float, max_iters: int, result: torch.Tensor,
                     F: torch.Tensor, mask: torch.Tensor, workspace: Optional[torch.Tensor] = None):
    """cv2.findFundamentalMat's small-input branches for the pairs of a batch with 7..14 matches (LMedS for 8..14,
    all 7-point solutions for exactly 7), written INTO result (B,4) / F (B,9) / mask (B,N) -- rows of other pairs
    stay as they are.  -> models7 (B,3,9) f64, nmodels7 (B,) i32 (non-zero only where n_match == 7)."""
    _lib.require_device()
    lib = _lib.load()
    pts4 = _dev(pts4, torch.float32, "pts4"); n_match = _dev(n_match, torch.int32, "n_match")
    B, N = pts4.shape[0], pts4.shape[1]
    for t, shape, dt, name in ((result, (B, 4), torch.int32, "result"), (F, (B, 9), torch.float64, "F"), (mask, (B, N), torch.uint8, "mask")):
        if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
            raise _lib.VoError(f"fund_small_batch: {name} must be a contiguous device tensor {shape} {dt}")
    models7 = torch.zeros((B, 3, 9), dtype=torch.float64, device=pts4.device)
    nmodels7 = torch.empty((B,), dtype=torch.int32, device=pts4.device)
    need = lib.vo_fund_small_workspace_bytes(float(confidence), int(max_iters), B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need,), dtype=torch.uint8, device=pts4.device)
    _lib.check(lib.vo_fund_small_batch(_ptr(pts4), N, _ptr(n_match), B, float(confidence), int(max_iters), _ptr(result), _ptr(F),
                                       _ptr(mask), _ptr(models7), _ptr(nmodels7), _ptr(workspace),
                                       workspace.numel() * workspace.element_size(), _stream()), "vo_fund_small_batch")
    return models7, nmodels7


def match_compact(idx: torch.Tensor, keep: torch.Tensor, kp_q: torch.Tensor, kp_t: torch.Tensor,
                  xyz_q: Optional[torch.Tensor] = None, want_pts4: bool = True):
    """idx (B,nq,2) i32, keep (B,nq) u8, kp_q (B,nq,2) f32, kp_t (B,nt,2) f32, xyz_q (B,nq,3) f32 | None
    -> dict(match_q, match_t (B,nq) i32, n (B,) i32, pts4 (B,nq,4) f32 | None, obj (B,nq,3), img (B,nq,2) | None);
    rows beyond n[b] are unspecified."""
    _lib.require_device()
    idx = _dev(idx, torch.int32, "idx"); keep = _dev(keep, torch.uint8, "keep")
    kp_q = _dev(kp_q, torch.float32, "kp_q"); kp_t = _dev(kp_t, torch.float32, "kp_t")
    B, nq, nt = idx.shape[0], idx.shape[1], kp_t.shape[1]
    if tuple(keep.shape) != (B, nq) or tuple(kp_q.shape) != (B, nq, 2) or kp_t.shape[0] != B:
        raise _lib.VoError("match_compact: shape mismatch")
    dev = idx.device
    mq = torch.empty((B, nq), dtype=torch.int32, device=dev); mt = torch.empty_like(mq)
    n = torch.empty((B,), dtype=torch.int32, device=dev)
    pts4 = torch.empty((B, nq, 4), dtype=torch.float32, device=dev) if want_pts4 else None
    obj = img = None
    if xyz_q is not None:
        xyz_q = _dev(xyz_q, torch.float32, "xyz_q")
        obj = torch.empty((B, nq, 3), dtype=torch.float32, device=dev)
        img = torch.empty((B, nq, 2), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().vo_match_compact(_ptr(idx), _ptr(keep), nq, nt, B, _ptr(kp_q), _ptr(kp_t), _ptr(xyz_q), _ptr(mq),
                                            _ptr(mt), _ptr(n), _ptr(pts4), _ptr(obj), _ptr(img), _stream()), "vo_match_compact")
    return dict(match_q=mq, match_t=mt, n=n, pts4=pts4, obj=obj, img=img)


def gen_subsets_device(n_dev: Optional[torch.Tensor], N: int, model_points: int, H: int,
                       pts4: Optional[torch.Tensor] = None, batch: Optional[int] = None):
    """cv::RNG + getSubset on the device -> samples (B,H,model_points) i32 (-1 rows = none), n_generated (B,) i32."""
    _lib.require_device()
    if n_dev is not None:
        n_dev = _dev(n_dev, torch.int32, "n_dev"); B = n_dev.numel(); dev = n_dev.device
    else:
        B = pts4.shape[0] if pts4 is not None else int(batch); dev = pts4.device if pts4 is not None else torch.device("cuda", torch.cuda.current_device())
    if pts4 is not None:
        pts4 = _dev(pts4, torch.float32, "pts4")
        if pts4.shape[0] != B or pts4.shape[1] != N or pts4.shape[2] != 4:
            raise _lib.VoError("gen_subsets_device: pts4 must be (B,N,4)")
    samples = torch.empty((B, H, model_points), dtype=torch.int32, device=dev)
    ngen = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().vo_gen_subsets_device(_ptr(n_dev), int(N), int(model_points), int(H), _ptr(pts4), B, _ptr(samples),
                                                 _ptr(ngen), _stream()), "vo_gen_subsets_device")
    return samples, ngen


# ------------------------------------------------------------------------------- host helpers
def gen_subsets(n_points: int, model_points: int, max_iters: int, pts4_host: Optional[np.ndarray] = None,
                max_attempts: int = 10000) -> np.ndarray:
    """cv::RNG((uint64)-1) + getSubset sample stream (host only). (n_generated, model_points) i32.
    max_attempts: 10000 in OpenCV's RANSAC registrator, 1000 (getSubset's default) in its LMedS registrator."""
    lib = _lib.load()
    out = np.zeros((max(int(max_iters), 1), model_points), dtype=np.int32)
    p = None
    if pts4_host is not None:
        pts4_host = np.ascontiguousarray(pts4_host, dtype=np.float32).reshape(-1, 4)
        p = pts4_host.ctypes.data
    n = lib.vo_host_gen_subsets_attempts(int(n_points), int(model_points), int(max_iters), int(max_attempts), p, out.ctypes.data)
    _lib.check(n, "vo_host_gen_subsets_attempts")
    return out[:n]


def _c12(P) -> "C.Array":
    import ctypes as C
    import numpy as np
    a = np.ascontiguousarray(np.asarray(P, dtype=np.float64)).reshape(-1)
    return (C.c_double * a.size)(*a.tolist())


def triangulate(P1, P2, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """cv.triangulatePoints on the device: P1, P2 (3,4) host; x1, x2 (N,2) f64 -> (N,4) f64 homogeneous."""
    _lib.require_device()
    x1 = _dev(x1, torch.float64, "x1").reshape(-1, 2)
    x2 = _dev(x2, torch.float64, "x2").reshape(-1, 2)
    if x1.shape != x2.shape:
        raise _lib.VoError("triangulate: point sets differ in length")
    X = torch.empty((x1.shape[0], 4), dtype=torch.float64, device=x1.device)
    _lib.check(_lib.load().vo_triangulate(_c12(P1), _c12(P2), _ptr(x1), _ptr(x2), x1.shape[0], _ptr(X), _stream()),
               "vo_triangulate")
    return X


def pose_tail_batch(F: torch.Tensor, mask: torch.Tensor, pts4: torch.Tensor, K, max_depth: float,
                    result: Optional[torch.Tensor] = None, n_dev: Optional[torch.Tensor] = None):
    """F (B,9) f64, mask (B,N) u8, pts4 (B,N,4) f32, K (3,3) host -> dict(cand (B,4,12), score (B,4) i32,
    T (B,4,4), X (B,N,4), keep (B,N) u8); see vo_pose_tail_batch in include/vo_hip.h."""
    _lib.require_device()
    F = _dev(F, torch.float64, "F").reshape(-1, 9)
    mask = _dev(mask, torch.uint8, "mask"); pts4 = _dev(pts4, torch.float32, "pts4")
    B, N = mask.shape
    if F.shape[0] != B or tuple(pts4.shape) != (B, N, 4):
        raise _lib.VoError("pose_tail_batch: shape mismatch")
    if result is not None:
        result = _dev(result, torch.int32, "result")
    if n_dev is not None:
        n_dev = _dev(n_dev, torch.int32, "n_dev")
    dev = F.device
    cand = torch.empty((B, 4, 12), dtype=torch.float64, device=dev)
    score = torch.empty((B, 4), dtype=torch.int32, device=dev)
    T = torch.empty((B, 4, 4), dtype=torch.float64, device=dev)
    X = torch.empty((B, N, 4), dtype=torch.float64, device=dev)
    keep = torch.empty((B, N), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().vo_pose_tail_batch(_ptr(F), _ptr(result), _ptr(mask), _ptr(pts4), _ptr(n_dev), _c12(K), N, B,
                                              float(max_depth), _ptr(cand), _ptr(score), _ptr(T), _ptr(X), _ptr(keep),
                                              _stream()), "vo_pose_tail_batch")
    return dict(cand=cand, score=score, T=T, X=X, keep=keep)


def epnp_batch(obj: torch.Tensor, img: torch.Tensor, mask: torch.Tensor, K, n_dev: Optional[torch.Tensor] = None,
               result: Optional[torch.Tensor] = None):
    """EPnP over the masked correspondences of every pair: obj (B,N,3) f32, img (B,N,2) f32, mask (B,N) u8
    -> Rt (B,12) f64 [R row-major | t], ok (B,) i32."""
    _lib.require_device()
    obj = _dev(obj, torch.float32, "obj"); img = _dev(img, torch.float32, "img"); mask = _dev(mask, torch.uint8, "mask")
    B, N = mask.shape
    if tuple(obj.shape) != (B, N, 3) or tuple(img.shape) != (B, N, 2):
        raise _lib.VoError("epnp_batch: shape mismatch")
    if n_dev is not None:
        n_dev = _dev(n_dev, torch.int32, "n_dev")
    if result is not None:
        result = _dev(result, torch.int32, "result")
    Rt = torch.empty((B, 12), dtype=torch.float64, device=mask.device)
    ok = torch.empty((B,), dtype=torch.int32, device=mask.device)
    _lib.check(_lib.load().vo_epnp_batch(_ptr(obj), _ptr(img), _ptr(mask), _ptr(n_dev), _ptr(result), _K4(K), N, B,
                                         _ptr(Rt), _ptr(ok), _stream()), "vo_epnp_batch")
    return Rt, ok


def lmeds_median_fund(models: torch.Tensor, pts4: torch.Tensor, valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models (H,9) f64, pts4 (N<=64,4) f32 -> (H,) f32 medians of the epipolar errors (rank N//2; +Inf where
    valid is 0): the ranking of OpenCV's LMedS registrator."""
    _lib.require_device()
    models = _dev(models, torch.float64, "models").reshape(-1, 9)
    pts4 = _dev(pts4, torch.float32, "pts4").reshape(-1, 4)
    H = models.shape[0]
    if valid is not None:
        valid = _dev(valid, torch.uint8, "valid").reshape(-1)
        if valid.numel() != H:
            raise _lib.VoError("lmeds_median_fund: valid must have one flag per model")
    med = torch.empty((H,), dtype=torch.float32, device=models.device)
    _lib.check(_lib.load().vo_lmeds_median_fund(_ptr(models), _ptr(valid), H, _ptr(pts4), pts4.shape[0], _ptr(med),
                                                _stream()), "vo_lmeds_median_fund")
    return med


def update_num_iters(p: float, ep: float, model_points: int, max_iters: int) -> int:
    """OpenCV's RANSACUpdateNumIters (host helper of the library)."""
    return int(_lib.load().vo_host_update_num_iters(float(p), float(ep), int(model_points), int(max_iters)))


# ---- synthetic inputs (sweep.py's generators; no counterpart in the reference) ---------------------------
def synth_sift_like(keys: torch.Tensor, n: int, stream_id: int) -> torch.Tensor:
    """(B,) int64 pair keys -> (B,n,128) f32 SIFT-like integer rows; value = f(key, stream_id, element index)."""
    lib = _lib.load()
    _lib.require_device()
    keys = _dev(keys, torch.int64, "keys")
    out = torch.empty((keys.numel(), int(n), 128), dtype=torch.float32, device=keys.device)
    _lib.check(lib.vo_synth_sift_like(_ptr(keys), keys.numel(), int(n), int(stream_id), _ptr(out), _stream()), "vo_synth_sift_like")
    return out


def synth_hash64(keys: torch.Tensor, n: int, stream_id: int) -> torch.Tensor:
    """(B,) int64 pair keys -> (B,n) int64 hash words (sweep.PairRng.bits in one launch)."""
    lib = _lib.load()
    _lib.require_device()
    keys = _dev(keys, torch.int64, "keys")
    out = torch.empty((keys.numel(), int(n)), dtype=torch.int64, device=keys.device)
    _lib.check(lib.vo_synth_hash64(_ptr(keys), keys.numel(), int(n), int(stream_id), _ptr(out), _stream()), "vo_synth_hash64")
    return out


def synth_noisy_copies(keys: torch.Tensor, query: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, stream_id: int,
                       sigma: float, train: torch.Tensor) -> torch.Tensor:
    """train[b, dst[b,m]] = saturate(round(query[b, src[b,m]] + sigma N(0,1))) in place; src/dst (B,nmatch) int64."""
    lib = _lib.load()
    _lib.require_device()
    keys, query, train = _dev(keys, torch.int64, "keys"), _dev(query, torch.float32, "query"), _dev(train, torch.float32, "train")
    src, dst = _dev(src, torch.int64, "src"), _dev(dst, torch.int64, "dst")
    B = keys.numel()
    if query.ndim != 3 or train.ndim != 3 or query.shape[0] != B or train.shape[0] != B or query.shape[2] != 128 or \
            train.shape[2] != 128 or src.shape != dst.shape or src.shape[0] != B:
        raise _lib.VoError("synth_noisy_copies: query (B,nq,128), train (B,nt,128), src/dst (B,nmatch)")
    _lib.check(lib.vo_synth_noisy_copies(_ptr(keys), B, _ptr(query), query.shape[1], _ptr(src), _ptr(dst), src.shape[1],
                                         int(stream_id), float(sigma), _ptr(train), train.shape[1], _stream()), "vo_synth_noisy_copies")
    return train
