#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X.

Workload (BASELINE.json configs[3], "cfg4"): per frame pair, 8192 x 8192 SIFT-like descriptors are
matched (f32 -> int8 conversion, exact 2-NN on int8 MFMA, ratio test) and two 4096-iteration RANSAC
runs are evaluated as batches: 7-point fundamental matrix (up to 3 models per sample, symmetric
epipolar error) and P3P PnP (reprojection error), each = minimal solves -> scoring of every model
against every correspondence -> replay of OpenCV's sequential best-model bookkeeping -> inlier mask
of the winner.  One "step" = one pass over `--pairs` frame pairs per GPU.  Inputs are synthetic,
seeded, and resident in HBM before the timed region; the sample index sets come from the restated
cv::RNG stream and are generated once on the host (they depend only on the point count).

Multi-GPU: frame pairs are independent, so every rank processes its own `--pairs` pairs and there
is no collective on the data path (weak scaling); value = pairs of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from visual_odometry_amd import _lib, ops, synth  # noqa: E402

NQ = NT = 8192
H = 4096                     # RANSAC iterations (sample sets) per problem
NPTS = 4096                  # correspondences per RANSAC problem (~ratio-test survivors of 8192)
F_THR, F_CONF = 0.1, 0.9999  # reference src/params.py:24-27
P_THR, P_CONF = 1.0, 0.9999  # reference src/params.py:72-75

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E
I8_OPS_PER_CLK_CU = 8192     # 4 SIMD x 2048 int8 ops/clk (v_mfma_i32_32x32x32_i8: 65536 ops / 32 clk)
FP64_PEAK_TFLOPS = 78.6      # 256 CU x 128 flop/clk x 2.4 GHz


class PairSet:
    """Device-resident inputs of `pairs` frame pairs (a few distinct seeds, tiled)."""

    def __init__(self, pairs: int, dev: torch.device, rank: int, distinct: int = 4):
        distinct = min(distinct, pairs)
        qs, ts, f_pts, f_sub, p_obj, p_img, p_sub = [], [], [], [], [], [], []
        for d in range(distinct):
            seed = 1000 * rank + d
            q, t = synth.sift_like_pair(NQ, NT, seed=1234 + seed)
            qs.append(q); ts.append(t)
            sc = synth.two_view_scene(NPTS, K=synth.K_KITTI, size=synth.IMAGE_SIZE["kitti"], seed=4321 + seed)
            pts4 = np.ascontiguousarray(np.concatenate([sc["p1"], sc["p2"]], axis=1), dtype=np.float32)
            f_pts.append(pts4)
            f_sub.append(ops.gen_subsets(NPTS, 7, H, pts4))
            ps = synth.pnp_scene(NPTS, K=synth.K_MALAGA, size=synth.IMAGE_SIZE["malaga"], seed=8765 + seed)
            p_obj.append(ps["obj"]); p_img.append(ps["img"])
            p_sub.append(ops.gen_subsets(NPTS, 4, H, None))
        sel = [i % distinct for i in range(pairs)]
        up = lambda lst: [torch.from_numpy(np.ascontiguousarray(lst[i])).to(dev) for i in sel]
        self.pairs = pairs
        self.q = torch.stack(up(qs))            # (B, NQ, 128) f32
        self.t = torch.stack(up(ts))
        self.f_pts, self.f_sub = up(f_pts), up(f_sub)
        self.p_obj, self.p_img, self.p_sub = up(p_obj), up(p_img), up(p_sub)
        self.ws = torch.empty((max(16, _lib.load().vo_knn2_workspace_bytes(NQ, NT, pairs)),), dtype=torch.uint8, device=dev)
        self.arange3 = torch.arange(3, device=dev, dtype=torch.int32)[None, :]


def step(ps: PairSet, ev=None):
    """One pass over all pairs.  `ev`, when given, collects (name, start_event, end_event)."""
    B = ps.pairs

    def timed(name, fn):
        if ev is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        ev.append((name, e0, e1))
        return r

    # ---- matching, all pairs in one launch each
    qi8, qn, _ = ops.desc_to_i8(ps.q.reshape(B * NQ, 128))
    ti8, tn, _ = ops.desc_to_i8(ps.t.reshape(B * NT, 128))
    idx, d2 = timed("knn2", lambda: ops.knn2(qi8.reshape(B, NQ, 128), qn.reshape(B, NQ), ti8.reshape(B, NT, 128),
                                             tn.reshape(B, NT), ps.ws))
    keep, nkeep = ops.ratio_test(d2.reshape(B * NQ, 2), 0.8)

    # ---- RANSAC, one problem per pair
    out = []
    for b in range(B):
        models, nm = ops.solve_7pt(ps.f_sub[b], ps.f_pts[b])
        valid = (ps.arange3 < nm[:, None]).to(torch.uint8)
        counts = timed("score_fund", lambda: ops.score_fund(models.reshape(-1, 9), ps.f_pts[b], F_THR, valid.reshape(-1)))
        res_f = ops.ransac_replay(counts.reshape(-1, 3), nm, NPTS, 7, F_CONF, H)
        best = models.reshape(-1, 9)[(res_f[0] * 3 + res_f[1]).clamp(min=0).long()]
        mask_f = ops.mask_fund(best, ps.f_pts[b], F_THR)

        pm, rv, pnm = ops.solve_p3p(ps.p_sub[b], ps.p_obj[b], ps.p_img[b], synth.K_MALAGA)
        pcounts = timed("score_pnp", lambda: ops.score_pnp(pm, ps.p_obj[b], ps.p_img[b], synth.K_MALAGA, P_THR,
                                                           pnm.to(torch.uint8)))
        res_p = ops.ransac_replay(pcounts, pnm, NPTS, 4, P_CONF, H)
        bestp = pm[res_p[0].clamp(min=0).long()]
        mask_p = ops.mask_pnp(bestp, ps.p_obj[b], ps.p_img[b], synth.K_MALAGA, P_THR)
        out.append((res_f, mask_f, res_p, mask_p))
    return idx, keep, nkeep, out


def cpu_baseline(max_seconds: float = 25.0):
    """The oracle (CPU restatement of the OpenCV routines; OpenCV itself is absent) on the same
    workload, a bounded number of frame pairs."""
    import oracle

    oracle.build()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    t0 = time.perf_counter()
    done = 0
    while done < 3 and (done == 0 or (time.perf_counter() - t0) * (done + 1) / done < max_seconds):
        q, t = synth.sift_like_pair(NQ, NT, seed=1234 + done)
        idx, dist = oracle.knn2(q, t, mt=True)
        oracle.ratio_test(dist, 0.8)
        sc = synth.two_view_scene(NPTS, K=synth.K_KITTI, size=synth.IMAGE_SIZE["kitti"], seed=4321 + done)
        oracle.find_fundamental_mat(sc["p1"], sc["p2"], F_THR, F_CONF, H)
        pp = synth.pnp_scene(NPTS, K=synth.K_MALAGA, size=synth.IMAGE_SIZE["malaga"], seed=8765 + done)
        oracle.solve_pnp_ransac(pp["obj"], pp["img"], pp["K"], H, P_THR, P_CONF)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame pairs of the same workload incl. input generation; matching on {threads} "
                      f"OpenMP threads, RANSAC loops serial as in OpenCV; CPU restatement, not OpenCV"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=16, help="frame pairs per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.require_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    ps = PairSet(args.pairs, dev, rank)
    for _ in range(args.warmup):
        step(ps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(ps, ev)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-kernel durations from the HIP events recorded inside the timed region
    dur = {}
    for name, e0, e1 in ev:
        dur.setdefault(name, []).append(e0.elapsed_time(e1) * 1e-3)
    avg = {k: float(np.mean(v)) for k, v in dur.items()}
    tot = {k: float(np.sum(v)) for k, v in dur.items()}

    prop = torch.cuda.get_device_properties(dev)
    cus = prop.multi_processor_count
    clock_ghz = 2.4
    i8_peak_tops = cus * I8_OPS_PER_CLK_CU * clock_ghz * 1e9 / 1e12

    # score_fund: one launch = (3*H models) x NPTS correspondences; algorithmic bytes per evaluation
    # = 16 B (one x1 y1 x2 y2 f32 correspondence) + per model 72 B read + 4 B count
    sf_bytes = 3 * H * NPTS * 16 + 3 * H * (72 + 4)
    sf_gbs = sf_bytes / avg["score_fund"] / 1e9
    sp_bytes = H * NPTS * 20 + H * (96 + 4)
    sp_gbs = sp_bytes / avg["score_pnp"] / 1e9
    knn_ops = 2.0 * NQ * NT * 128 * args.pairs
    knn_tops = knn_ops / avg["knn2"] / 1e12

    dominant = max(tot, key=tot.get)
    roof = {
        "score_fund": {"kernel": "score_fund_kernel", "bound": "hbm", "achieved": sf_gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": sf_gbs / HBM_PEAK_GBS, "traffic": None,
                       "note": "effective streaming bytes (16 B per model x correspondence); the tiled kernel "
                               "re-uses points from LDS so physical HBM traffic is far lower and the real limiter "
                               "is the f64 VALU",
                       "f64_tflops": 3 * H * NPTS * 49 / avg["score_fund"] / 1e12, "f64_peak_tflops": FP64_PEAK_TFLOPS},
        "score_pnp": {"kernel": "score_pnp_kernel", "bound": "hbm", "achieved": sp_gbs, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": sp_gbs / HBM_PEAK_GBS, "traffic": None},
        "knn2": {"kernel": "knn2_partial_kernel", "bound": "mfma", "achieved": knn_tops, "peak": i8_peak_tops,
                 "unit": "TOP/s", "frac": knn_tops / i8_peak_tops, "traffic": None,
                 "note": f"int8 dense peak = {cus} CU x {I8_OPS_PER_CLK_CU} ops/clk x {clock_ghz} GHz"},
    }
    pairs_total = args.pairs * world * args.steps
    evals_per_step = args.pairs * (3 * H * NPTS + H * NPTS)
    line = {
        "metric": "matched frame-pairs/sec", "value": pairs_total / dt, "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8 (matching) / f64 (RANSAC)",
        "data": "synthetic",
        "config": {"workload": "cfg4: 8192x8192x128 SIFT match + ratio test, then 4096-iteration 7-point F-RANSAC "
                               "(12288 model slots x 4096 pts) and 4096-iteration P3P PnP-RANSAC (4096 x 4096 pts) per frame pair",
                   "pairs_per_gpu_per_step": args.pairs, "parallelism": f"pairs sharded over {world} GPU(s), no collective"},
        "ransac_scoring": {"metric": "RANSAC hypotheses x points scored/sec",
                           "value": (3 * H * NPTS) / avg["score_fund"],
                           "unit": "evals/s per GPU inside score_fund_kernel",
                           "pnp_value": (H * NPTS) / avg["score_pnp"],
                           "whole_job_evals_per_s": evals_per_step * world * args.steps / dt},
        "kernel_time_share": {k: v / dt for k, v in tot.items()},
        "kernel_avg_us": {k: v * 1e6 for k, v in avg.items()},
        "roofline": roof[dominant],
        "roofline_other": [roof[k] for k in roof if k != dominant],
        "device": prop.name, "cus": cus,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
