"""Frame pairs end to end on the device: descriptors and keypoint positions in,
match lists, F-RANSAC and PnP-RANSAC results out, with no host round trip in between.  This is the
batched form of the reference chain  match_features -> get_positions -> calc_fundamental_mat
(src/twoDtwoD.py:49-57)  and  match_features -> pnp  (src/sift_test.py:104-119, src/pnp.py:13-41)
for many pairs of one shape; the single-pair, host-array entry points live in ``features``,
``utils.geometry`` and ``pnp``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops


@dataclass
class PairBatchResult:
    idx: torch.Tensor            # (B,nq,2) i32   two nearest train rows of every query row
    d2: torch.Tensor             # (B,nq,2) i32   their exact squared distances
    match_q: torch.Tensor        # (B,nq) i32     query rows that pass the ratio test, ascending; first n_match used
    match_t: torch.Tensor        # (B,nq) i32     their matches
    n_match: torch.Tensor        # (B,) i32
    pts4: torch.Tensor           # (B,nq,4) f32   matched positions x_q y_q x_t y_t
    f_result: torch.Tensor       # (B,4) i32      best iteration, best model, inliers, final niters
    F: torch.Tensor              # (B,9) f64
    f_mask: torch.Tensor         # (B,nq) u8      over the matched correspondences
    p_result: Optional[torch.Tensor] = None
    pose: Optional[torch.Tensor] = None          # (B,12) f64 [R|t] of the best P3P model (before the EPnP refit)
    rvec: Optional[torch.Tensor] = None          # (B,3) f64
    p_mask: Optional[torch.Tensor] = None        # (B,nq) u8
    pose_refined: Optional[torch.Tensor] = None  # (B,12) f64 [R|t] of the EPnP refit over the PnP inliers: what cv.solvePnPRansac returns
    p_ok: Optional[torch.Tensor] = None          # (B,) i32 1 where pose_refined holds a pose
    pnp_done: Optional["torch.cuda.Event"] = None  # recorded behind the PnP chain (see wait_pnp)

    def wait_pnp(self) -> None:
        """Make the current stream wait for the PnP results (only needed after wait_pnp=False)."""
        if self.pnp_done is not None:
            torch.cuda.current_stream().wait_event(self.pnp_done)
    f_samples: Optional[torch.Tensor] = None     # (B,f_iters,7) i32 the sample streams used
    p_samples: Optional[torch.Tensor] = None
    obj: Optional[torch.Tensor] = None           # (B,nq,3) f32 / (B,nq,2) f32: the PnP correspondences
    img: Optional[torch.Tensor] = None
    f7_models: Optional[torch.Tensor] = None     # (B,3,9) f64 every 7-point solution of the pairs with EXACTLY 7 matches (cv2 stacks them)
    f7_count: Optional[torch.Tensor] = None      # (B,) i32 how many (0 for every other pair)
    e_result: Optional[torch.Tensor] = None      # (B,4) i32  the essential-matrix chain (e_iters > 0): as f_result
    E: Optional[torch.Tensor] = None             # (B,9) f64  best five-point model
    e_mask: Optional[torch.Tensor] = None        # (B,nq) u8  its Sampson inliers over the matched correspondences
    e_samples: Optional[torch.Tensor] = None     # (B,e_iters,5) i32
    pts4n: Optional[torch.Tensor] = None         # (B,nq,4) f64 the K-normalised correspondences the E chain ran on
    desc_flags: Optional[torch.Tensor] = None    # (2,) i32 VO_DESC_BAD_* bits of the query / train descriptors: callers
                                                 # MUST check it (check_descriptors) where they synchronise anyway --
                                                 # the int8 matcher is only exact for integer 0..255 rows of norm <= 1024

    def check_descriptors(self) -> None:
        """Raise VoError if the descriptor conversion flagged non-integer / out-of-range values or rows whose
        norm exceeds 1024 (one host synchronisation).  The single-pair entry points raise by themselves."""
        if self.desc_flags is None:
            return
        fq, ft = (int(v) for v in self.desc_flags.cpu())
        if fq | ft:
            what = []
            if (fq | ft) & ops._lib.VO_DESC_BAD_VALUE:
                what.append("descriptor values must be integers in [0, 255]")
            if (fq | ft) & ops._lib.VO_DESC_BAD_NORM:
                what.append("a descriptor row has L2 norm > 1024 (OpenCV's f32 sqrt keys could tie where the integer keys do not)")
            raise ops._lib.VoError(f"batched matching: {'; '.join(what)} (query flags {fq}, train flags {ft}): "
                                   "results would differ from cv.BFMatcher")


@dataclass
class PreparedDescriptors:
    """A batch's descriptors already converted for the matcher (prepare_descriptors): lets a caller that walks
    through many batches run the HBM-bound conversion of the NEXT batch underneath the VALU-bound scoring of the
    current one."""
    qi8: torch.Tensor            # (B*nq,128) i8
    qn: torch.Tensor             # (B*nq,) i32
    ti8: torch.Tensor
    tn: torch.Tensor
    flags: torch.Tensor          # (2,) i32 VO_DESC_BAD_* bits
    shape: tuple                 # (B, nq, nt)
    ready: "torch.cuda.Event"    # recorded behind the conversion
    consumed: Optional["torch.cuda.Event"] = None   # recorded by match_and_ransac behind the kNN pass that read it


_side = {}
_SIDE_MAX = 16


def _side_stream(dev: torch.device, main: "torch.cuda.Stream", role: str = "pnp") -> "torch.cuda.Stream":
    """The PnP chain's stream (or the descriptor prefetch stream): one per (device, caller's stream, role), so that
    callers that run several batches concurrently on streams of their own do not serialise their side work on one."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), main.cuda_stream, role)
    if key not in _side:
        while len(_side) >= _SIDE_MAX:                 # callers that keep creating streams: drop the oldest entry
            _side.pop(next(iter(_side)))
        # default priority on purpose: a high-priority stream changed nothing in the measured overlap, and
        # with an RCCL communicator alive in the process (bench.py under torch.distributed.run) the
        # cross-stream waits of the chain never completed on it
        import os
        prio = int(os.environ.get("VO_SIDE_PRIORITY", "0"))      # diagnostic knob only (scripts/diag/dist_hang.py)
        _side[key] = torch.cuda.Stream(device=dev, priority=prio)
    return _side[key]


def prepare_descriptors(desc_q: torch.Tensor, desc_t: torch.Tensor, reuse: Optional[PreparedDescriptors] = None) -> PreparedDescriptors:
    """Enqueue the int8 conversion of a batch's descriptors ((B,n,128) f32, resident) on the pipeline's prefetch
    stream and return at once; pass the result to match_and_ransac(prepared=...).  `reuse`: a PreparedDescriptors of
    the same shape whose buffers may be overwritten (the conversion waits until the matcher that read them is done) --
    two of them, used alternately, pipeline a stream of batches without allocating."""
    B, nq, nt = desc_q.shape[0], desc_q.shape[1], desc_t.shape[1]
    main = torch.cuda.current_stream()
    pre = _side_stream(desc_q.device, main, "prefetch")
    pre.wait_stream(main)                                  # the inputs were produced on the caller's stream
    if reuse is not None:
        if reuse.shape != (B, nq, nt):
            raise ops._lib.VoError(f"prepare_descriptors: reuse has shape {reuse.shape}, batch has {(B, nq, nt)}")
        if reuse.consumed is not None:
            pre.wait_event(reuse.consumed)
    for t in (desc_q, desc_t):
        t.record_stream(pre)
    with torch.cuda.stream(pre):
        if reuse is not None:
            flags = reuse.flags.zero_()
            qi8, qn, _ = ops.desc_to_i8(desc_q.reshape(B * nq, 128), flags[0:1], out=(reuse.qi8, reuse.qn))
            ti8, tn, _ = ops.desc_to_i8(desc_t.reshape(B * nt, 128), flags[1:2], out=(reuse.ti8, reuse.tn))
        else:
            flags = torch.zeros((2,), dtype=torch.int32, device=desc_q.device)
            qi8, qn, _ = ops.desc_to_i8(desc_q.reshape(B * nq, 128), flags[0:1])
            ti8, tn, _ = ops.desc_to_i8(desc_t.reshape(B * nt, 128), flags[1:2])
        ready = pre.record_event()
    for t in (qi8, qn, ti8, tn, flags):
        t.record_stream(main)
    return PreparedDescriptors(qi8, qn, ti8, tn, flags, (B, nq, nt), ready)


def workspace_bytes(nq: int, nt: int, batch: int, f_iters: int, p_iters: int = 0) -> int:
    """Bytes of scratch `match_and_ransac` wants in one buffer: the kNN pass and the two RANSAC runs, which
    may all overlap in time (PnP of one call with the kNN of the next when wait_pnp=False), so they get
    disjoint parts."""
    lib = ops._lib.load()
    r = _ws_align(lib.vo_ransac_workspace_bytes(ops.KIND_FUND, f_iters, batch))
    if p_iters > 0:
        r += _ws_align(lib.vo_ransac_workspace_bytes(ops.KIND_PNP, p_iters, batch))
    return _ws_align(max(16, lib.vo_knn2_workspace_bytes(nq, nt, batch))) + r


def _ws_align(nbytes: int) -> int:
    return (nbytes + 255) // 256 * 256


def _split_workspace(workspace, nq, nt, f_iters, p_iters, batch):
    """Disjoint parts of the caller's buffer for the kNN pass, the F and the PnP run, or Nones when it is
    absent or too small (each stage then allocates its own)."""
    if workspace is None:
        return None, None, None
    lib = ops._lib.load()
    nf = _ws_align(lib.vo_ransac_workspace_bytes(ops.KIND_FUND, f_iters, batch))
    np_ = _ws_align(lib.vo_ransac_workspace_bytes(ops.KIND_PNP, p_iters, batch)) if p_iters > 0 else 0
    nk = _ws_align(max(16, lib.vo_knn2_workspace_bytes(nq, nt, batch)))
    flat = workspace.reshape(-1).view(torch.uint8)
    if flat.numel() < nk + nf + np_:
        return None, None, None
    return flat[:nk], flat[nk:nk + nf], (flat[nk + nf:nk + nf + np_] if np_ else None)


def _small_pairs(pts4, n, f_confidence, f_iters, f_res, F, f_mask, ws_f):
    """Pairs with 7..14 matches, where cv2.findFundamentalMat does not run 7-point RANSAC: LMedS for 8..14, the plain
    7-point solutions for exactly 7 (ops.fund_small_batch), written into the F batch's outputs on the same stream.
    The F run's workspace is free again by then and is re-used when it is large enough."""
    need = ops._lib.load().vo_fund_small_workspace_bytes(float(f_confidence), int(f_iters), pts4.shape[0])
    ws = ws_f if ws_f is not None and ws_f.numel() >= need else None
    return ops.fund_small_batch(pts4, n, f_confidence, f_iters, f_res, F, f_mask, workspace=ws)


def match_and_ransac(desc_q: torch.Tensor, desc_t: torch.Tensor, kp_q: torch.Tensor, kp_t: torch.Tensor, ratio: float,
                     f_threshold: float, f_confidence: float, f_iters: int,
                     xyz_q: Optional[torch.Tensor] = None, K=None, p_threshold: float = 1.0, p_confidence: float = 0.9999,
                     p_iters: int = 0, first_block_f: int = 0, first_block_pnp: int = 0,
                     workspace: Optional[torch.Tensor] = None, wait_pnp: bool = True, group=None,
                     prepared: Optional[PreparedDescriptors] = None, e_iters: int = 0, e_threshold: float = 1.0,
                     e_confidence: float = 0.9999, e_max_iters: Optional[int] = None, first_block_e: int = 0) -> PairBatchResult:
    """desc_* (B,n,128) f32 integer-valued, kp_* (B,n,2) f32, xyz_q (B,nq,3) f32 landmarks of the query
    keypoints (enables the PnP run when p_iters > 0).  first_block_*: see ops.ransac_batch (0 = evaluate all
    iterations at once; first_block_f = 256 is what bench.py and sweep.run_sweep use: hypotheses that cannot beat the
    best count of the iterations before them get an upper bound instead of a count, same F / mask / record).  Everything is enqueued on the current stream, except the PnP chain (P3P RANSAC +
    EPnP refit), which runs on a side stream; with wait_pnp=True (default) the current
    stream waits for it before returning, with wait_pnp=False the caller does (`result.wait_pnp()`), which
    lets the small, latency-bound PnP kernels of one batch fill the gaps of the next batch's big ones.

    Per pair the RANSAC runs see exactly the correspondences, in the order, that the reference's
    gathers would hand to cv2 -- so results equal the single-pair entry points', including cv2's small-input
    branches of findFundamentalMat: LMedS for 8..14 matches, and for exactly 7 all solutions of the seven points
    (f7_models / f7_count; F holds the first).  Fewer than 7 (F) / 5 (PnP) matches: no model.

    prepared: the descriptors' int8 conversion done ahead of time on the prefetch stream (prepare_descriptors); the
    conversion is then not repeated here.

    e_iters > 0 (needs K): the essential-matrix chain of src/utils/geometry.py:43-45 as well -- cv2.findEssentialMat(p1, p2,
    K, RANSAC, e_confidence, e_threshold, e_max_iters) over the matches: five-point models of `e_iters` getSubset samples
    scored by the Sampson distance on K-normalised f64 coordinates, threshold / ((fx + fy) / 2).  e_max_iters is OpenCV's
    maxIters (the reference's default is 50000) and defaults to e_iters; the answer is OpenCV's as long as its adaptive
    loop ends within the e_iters samples evaluated (e_result[:, 3] <= e_iters -- check it where you synchronise anyway).
    f_iters = 0 skips the fundamental-matrix chain.

    group (parallel.WORLD or a process group): ONE batch over several GPUs -- every rank passes the same
    inputs and gets the same outputs; see match_and_ransac_sharded."""
    if group is not None:
        if e_iters > 0 or f_iters <= 0:
            raise ops._lib.VoError("match_and_ransac: the essential-matrix chain is not available together with `group`")
        if prepared is not None or first_block_f != 0:
            raise ops._lib.VoError("match_and_ransac: `prepared` and `first_block_f` are not supported together with `group` "
                                   "(the sharded path converts its own query slice and evaluates every F iteration); "
                                   "wait_pnp has no effect there either: everything runs on the current stream")
        return match_and_ransac_sharded(desc_q, desc_t, kp_q, kp_t, ratio, f_threshold, f_confidence, f_iters, xyz_q=xyz_q, K=K,
                                        p_threshold=p_threshold, p_confidence=p_confidence, p_iters=p_iters,
                                        workspace=workspace, group=group, first_block_pnp=first_block_pnp)
    B, nq, nt = desc_q.shape[0], desc_q.shape[1], desc_t.shape[1]
    if prepared is not None:
        if prepared.shape != (B, nq, nt):
            raise ops._lib.VoError(f"match_and_ransac: prepared descriptors have shape {prepared.shape}, batch has {(B, nq, nt)}")
        torch.cuda.current_stream().wait_event(prepared.ready)
        qi8, qn, ti8, tn = prepared.qi8, prepared.qn, prepared.ti8, prepared.tn
        # this batch's own copy of the flag word (8 bytes, on the caller's stream, behind `ready`): the prefetch
        # buffer's word is zeroed and OR-ed into again as soon as the buffer is reused for a later batch
        flags = prepared.flags.clone()
    else:
        flags = torch.zeros((2,), dtype=torch.int32, device=desc_q.device)       # one fill for both conversions
        qi8, qn, _ = ops.desc_to_i8(desc_q.reshape(B * nq, 128), flags[0:1])
        ti8, tn, _ = ops.desc_to_i8(desc_t.reshape(B * nt, 128), flags[1:2])
    do_pnp = p_iters > 0 and xyz_q is not None
    ws_k, ws_f, ws_p = _split_workspace(workspace, nq, nt, f_iters, p_iters if do_pnp else 0, B)
    idx, d2 = ops.knn2(qi8.reshape(B, nq, 128), qn.reshape(B, nq), ti8.reshape(B, nt, 128), tn.reshape(B, nt), ws_k)
    if prepared is not None:
        prepared.consumed = torch.cuda.current_stream().record_event()
    keep, _ = ops.ratio_test(d2.reshape(B * nq, 2), ratio)
    c = ops.match_compact(idx, keep.reshape(B, nq), kp_q, kp_t, xyz_q if p_iters > 0 else None)
    n = c["n"]
    # The PnP chain needs nothing from the F chain, and the F chain starts with the latency-bound
    # getSubset kernel (a few workgroups per pair): run PnP on a side stream underneath it.
    main = torch.cuda.current_stream()
    pnp_out = None
    if do_pnp:
        side = _side_stream(desc_q.device, main)
        side.wait_stream(main)
        for t in (c["obj"], c["img"], n):
            t.record_stream(side)                  # allocated on the caller's stream, read on the side stream
        with torch.cuda.stream(side):
            n_p = torch.where(n >= 5, n, torch.zeros_like(n))
            p_samples, _ = ops.gen_subsets_device(n_p, nq, 4, p_iters, None)
            p_res, pose, rvec, p_mask = ops.ransac_batch(ops.KIND_PNP, p_samples, c["obj"], p_threshold, p_confidence, p_iters,
                                                         img=c["img"], K=K, n_dev=n_p, workspace=ws_p, first_block=first_block_pnp)
            refined, p_ok = ops.epnp_batch(c["obj"], c["img"], p_mask, K, n_dev=n_p, result=p_res)
            pnp_out = (p_res, pose, rvec, p_mask, p_samples, refined, p_ok)
            for t in pnp_out + (n_p,):
                t.record_stream(main)              # allocated on the side stream, consumed on the caller's
    f_res = F = f_mask = f_samples = f7 = n7 = None
    if f_iters > 0:
        # fewer than 15 matches: OpenCV would not run 7-point RANSAC; feed an empty problem instead
        n_f = torch.where(n >= 15, n, torch.zeros_like(n))
        f_samples, _ = ops.gen_subsets_device(n_f, nq, 7, f_iters, c["pts4"])
        f_res, F, _, f_mask = ops.ransac_batch(ops.KIND_FUND, f_samples, c["pts4"], f_threshold, f_confidence, f_iters,
                                               n_dev=n_f, workspace=ws_f, first_block=first_block_f)
        f7, n7 = _small_pairs(c["pts4"], n, f_confidence, f_iters, f_res, F, f_mask, ws_f)
    out = PairBatchResult(idx, d2, c["match_q"], c["match_t"], n, c["pts4"], f_res, F, f_mask, f_samples=f_samples,
                          desc_flags=flags, f7_models=f7, f7_count=n7)
    if e_iters > 0:
        if K is None:
            raise ops._lib.VoError("match_and_ransac: the essential-matrix chain needs K")
        import numpy as _np
        Kh = _np.asarray(K, dtype=_np.float64)
        fx, fy, cx, cy = float(Kh[0, 0]), float(Kh[1, 1]), float(Kh[0, 2]), float(Kh[1, 2])
        ifx, ify = 1.0 / fx, 1.0 / fy
        # cv2.findEssentialMat's normalisation, the operations of ransac.find_essential_ransac: u * (1/fx) + (-cx * (1/fx))
        scale = torch.tensor([ifx, ify, ifx, ify], dtype=torch.float64, device=desc_q.device)
        shift = torch.tensor([-cx * ifx, -cy * ify, -cx * ifx, -cy * ify], dtype=torch.float64, device=desc_q.device)
        pts4n = c["pts4"].to(torch.float64) * scale
        pts4n = pts4n + shift
        # 6+ matches run the RANSAC registrator (exactly 5 would return every solution stacked: single-pair entry point only)
        n_e = torch.where(n >= 6, n, torch.zeros_like(n))
        e_samples, _ = ops.gen_subsets_device(n_e, nq, 5, e_iters, None)
        e_res, E, _, e_mask = ops.ransac_batch(ops.KIND_ESS, e_samples, pts4n, e_threshold / ((fx + fy) / 2), e_confidence,
                                               e_max_iters if e_max_iters else e_iters, n_dev=n_e, first_block=first_block_e)
        out.e_result, out.E, out.e_mask, out.e_samples, out.pts4n = e_res, E, e_mask, e_samples, pts4n
    if do_pnp:
        out.pnp_done = side.record_event()
        if wait_pnp:
            main.wait_event(out.pnp_done)
        out.p_result, out.pose, out.rvec, out.p_mask, out.p_samples, out.pose_refined, out.p_ok = pnp_out
        out.obj, out.img = c["obj"], c["img"]
    return out


def match_and_ransac_sharded(desc_q: torch.Tensor, desc_t: torch.Tensor, kp_q: torch.Tensor, kp_t: torch.Tensor, ratio: float,
                             f_threshold: float, f_confidence: float, f_iters: int,
                             xyz_q: Optional[torch.Tensor] = None, K=None, p_threshold: float = 1.0,
                             p_confidence: float = 0.9999, p_iters: int = 0, workspace: Optional[torch.Tensor] = None,
                             group="world", first_block_pnp: int = 0, pnp_split: str = "pairs") -> PairBatchResult:
    """match_and_ransac for ONE batch of frame pairs spread over the ranks of `group` (BASELINE
    cfg4 "sharded over 2/4/8 GPUs"): every rank holds the same inputs and returns the same PairBatchResult,
    bit for bit the unsharded one.
      * matching: the query rows of every pair are split over the ranks (train rows replicated); one
        all_gather of the (idx, d2) rows assembles the 2-NN table everywhere;
      * ratio test, compaction, getSubset: replicated (cheap, O(nq));
      * F-RANSAC: the iteration range of every problem is split over the ranks (parallel.ransac_batch_sharded:
        one all_gather of count tables + one all-reduce of the winners); all iterations are evaluated (at the
        reference's 0.1 px threshold OpenCV's loop runs them all anyway);
      * PnP-RANSAC + EPnP refit: pnp_split="pairs" (default) deals whole problems to the ranks, keeping the
        early-termination blocks (first_block_pnp), and gathers the packed results with one all_gather;
        pnp_split="iterations" splits every problem's iteration range like the F run.
    Everything runs on the current stream: with collectives on the path the PnP chain is not moved to a side
    stream."""
    from . import parallel as par
    rank, world = par.world_info(group)
    B, nq, nt = desc_q.shape[0], desc_q.shape[1], desc_t.shape[1]
    flags = torch.zeros((2,), dtype=torch.int32, device=desc_q.device)
    qlo, qhi = par.shard_bounds(nq, rank, world)
    nql = qhi - qlo
    qi8, qn, _ = ops.desc_to_i8(desc_q[:, qlo:qhi].reshape(B * nql, 128), flags[0:1])
    ti8, tn, _ = ops.desc_to_i8(desc_t.reshape(B * nt, 128), flags[1:2])
    do_pnp = p_iters > 0 and xyz_q is not None
    ws_k, ws_f, ws_p = _split_workspace(workspace, nq, nt, f_iters, p_iters if do_pnp else 0, B)
    idx_l, d2_l = ops.knn2(qi8.reshape(B, nql, 128), qn.reshape(B, nql), ti8.reshape(B, nt, 128), tn.reshape(B, nt), ws_k)
    both = par.all_gather_dim1(torch.cat([idx_l, d2_l], dim=2), nq, group)           # (B, nq, 4)
    idx, d2 = both[:, :, :2].contiguous(), both[:, :, 2:].contiguous()
    keep, _ = ops.ratio_test(d2.reshape(B * nq, 2), ratio)
    c = ops.match_compact(idx, keep.reshape(B, nq), kp_q, kp_t, xyz_q if p_iters > 0 else None)
    n = c["n"]
    n_f = torch.where(n >= 15, n, torch.zeros_like(n))
    f_samples, _ = ops.gen_subsets_device(n_f, nq, 7, f_iters, c["pts4"])
    f_res, F, _, f_mask = par.ransac_batch_sharded(ops.KIND_FUND, f_samples, c["pts4"], f_threshold, f_confidence, f_iters,
                                                   n_dev=n_f, workspace=ws_f, group=group)
    f7, n7 = _small_pairs(c["pts4"], n, f_confidence, f_iters, f_res, F, f_mask, ws_f)         # replicated: O(pairs) work
    out = PairBatchResult(idx, d2, c["match_q"], c["match_t"], n, c["pts4"], f_res, F, f_mask, f_samples=f_samples,
                          desc_flags=flags, f7_models=f7, f7_count=n7)
    if do_pnp:
        n_p = torch.where(n >= 5, n, torch.zeros_like(n))
        p_samples, _ = ops.gen_subsets_device(n_p, nq, 4, p_iters, None)
        if pnp_split == "iterations":
            p_res, pose, rvec, p_mask = par.ransac_batch_sharded(ops.KIND_PNP, p_samples, c["obj"], p_threshold, p_confidence,
                                                                 p_iters, img=c["img"], K=K, n_dev=n_p, workspace=ws_p, group=group)
            refined, p_ok = ops.epnp_batch(c["obj"], c["img"], p_mask, K, n_dev=n_p, result=p_res)
        else:
            # by pairs: OpenCV's adaptive loop usually leaves a PnP problem after a few iterations, so splitting its
            # iteration range buys nothing; every rank runs whole problems (with early-termination blocks) for its
            # share of the pairs and ONE all_gather of the packed per-pair results makes them resident everywhere
            blo, bhi = par.shard_bounds(B, rank, world)
            sl = slice(blo, bhi)
            if bhi > blo:
                r_l, pose_l, rvec_l, mask_l = ops.ransac_batch(ops.KIND_PNP, p_samples[sl], c["obj"][sl], p_threshold, p_confidence,
                                                               p_iters, img=c["img"][sl], K=K, n_dev=n_p[sl], workspace=ws_p,
                                                               first_block=first_block_pnp)
                ref_l, ok_l = ops.epnp_batch(c["obj"][sl], c["img"][sl], mask_l, K, n_dev=n_p[sl], result=r_l)
                u8 = lambda t: t.contiguous().view(torch.uint8).reshape(bhi - blo, -1)
                packed = torch.cat([u8(pose_l), u8(rvec_l), u8(ref_l), u8(r_l), u8(ok_l), mask_l], dim=1)
            else:
                packed = torch.empty((0, 27 * 8 + 5 * 4 + nq), dtype=torch.uint8, device=desc_q.device)
            full = par.all_gather_rows(packed, B, group) if par._live(group) else packed
            f64 = lambda a, b: full[:, a:b].contiguous().view(torch.float64)
            i32 = lambda a, b: full[:, a:b].contiguous().view(torch.int32)
            pose, rvec, refined = f64(0, 96), f64(96, 120), f64(120, 216)
            p_res, p_ok, p_mask = i32(216, 232), i32(232, 236).reshape(B), full[:, 236:].contiguous()
        out.p_result, out.pose, out.rvec, out.p_mask, out.p_samples, out.pose_refined, out.p_ok = (
            p_res, pose, rvec, p_mask, p_samples, refined, p_ok)
        out.obj, out.img = c["obj"], c["img"]
    return out


def finish_small_pairs(res: PairBatchResult, f_threshold: float, f_confidence: float, f_iters: int) -> int:
    """Host-side check point after a batch (one synchronisation): raises if the descriptor conversion flagged bad
    input (check_descriptors) and returns how many pairs went through cv2.findFundamentalMat's small-input branches
    (7..14 matches).  Those branches themselves now run inside the batch (ops.fund_small_batch: LMedS for 8..14,
    the plain 7-point solutions for 7); the arguments are kept for callers of the round-1 signature."""
    res.check_descriptors()
    n = res.n_match.cpu()
    return int(((n >= 7) & (n <= 14)).sum())


def pose_tail(res: PairBatchResult, K, max_depth: Optional[float] = None, wait: bool = True):
    """The 2D-2D pose tail of src/twoDtwoD.py:57-79 for every pair of `res`, on the device:
    E = K^T F K -> decomposeEssentialMatrix -> disambiguateRelativePose over the F inliers.
    -> dict(cand, score, T (B,4,4) = T_cami_camj, X (B,nq,4) points in camera-i coordinates, keep (B,nq) u8);
    rows follow res.pts4 (matched correspondences, first n_match[b] used).

    wait=False: the tail (f64 Jacobi SVDs, latency bound) runs on a side stream of the caller's stream and the
    caller's stream does NOT wait for it -- like the PnP chain with wait_pnp=False it then runs underneath the next
    batch's matching; the dict carries the event `done` (torch.cuda.current_stream().wait_event(tail["done"]))."""
    if max_depth is None:
        from .params import get_params
        max_depth = get_params().INIT_PARAMS.MAX_DEPTH_DISTANCE
    if wait:
        return ops.pose_tail_batch(res.F, res.f_mask, res.pts4, K, max_depth, result=res.f_result, n_dev=res.n_match)
    main = torch.cuda.current_stream()
    side = _side_stream(res.F.device, main, "tail")
    side.wait_stream(main)
    ins = (res.F, res.f_mask, res.pts4, res.f_result, res.n_match)
    for t in ins:
        t.record_stream(side)
    with torch.cuda.stream(side):
        out = ops.pose_tail_batch(res.F, res.f_mask, res.pts4, K, max_depth, result=res.f_result, n_dev=res.n_match)
        out["done"] = side.record_event()
    for t in out.values():
        if isinstance(t, torch.Tensor):
            t.record_stream(main)
    return out
