#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X, frame pairs end to end on the device.

Workload (BASELINE.json configs[3], "cfg4"): per frame pair, 8192 x 8192 SIFT-like descriptors are
matched (f32 -> int8 conversion, exact 2-NN on int8 MFMA, ratio test 0.8), the surviving matches
are compacted into correspondences (about 4.7k per pair), and two 4096-iteration RANSAC runs are
evaluated over them: 7-point fundamental matrix (threshold 0.1 px, confidence 0.9999: the
reference's F parameters; up to 3 models per sample, symmetric epipolar error) and P3P PnP
(1 px, 0.9999: the reference's Malaga parameters; reprojection error).  Each RANSAC run = sample
stream of cv::RNG/getSubset generated on the device for the pair's match count -> minimal solves ->
scoring of every model against every correspondence -> replay of OpenCV's sequential best-model
bookkeeping incl. its adaptive iteration count -> inlier mask of the winner.  The PnP run evaluates
its iterations in blocks (256, then 4x more) and stops where OpenCV's loop would have stopped.
One "step" = one pass over `--pairs` frame pairs per GPU with no host synchronisation inside.
Inputs are synthetic, seeded, and resident in HBM before the timed region.

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

from visual_odometry_amd import _lib, ops, pipeline, synth  # noqa: E402

NQ = NT = 8192
H = 4096                      # RANSAC iterations (sample sets) per problem
RATIO = 0.8                   # reference src/params.py:10
F_THR, F_CONF = 0.1, 0.9999   # reference src/params.py:24-27
P_THR, P_CONF = 1.0, 0.9999   # reference src/params.py:72-75
FIRST_BLOCK = 64              # PnP iterations evaluated before the first early-termination check (then x5 per block)
K = synth.K_MALAGA
SIZE = synth.IMAGE_SIZE["malaga"]

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s HBM3E
I8_OPS_PER_CLK_CU = 8192      # 4 SIMD x 2048 int8 ops/clk (v_mfma_i32_32x32x32_i8: 65536 ops / 32 clk)
FP64_PEAK_TFLOPS = 78.6       # 256 CU x 128 flop/clk x 2.4 GHz


class PairSet:
    """Device-resident inputs of `pairs` frame pairs (a few distinct seeds, tiled)."""

    def __init__(self, pairs: int, dev: torch.device, rank: int, distinct: int = 4):
        distinct = min(distinct, pairs)
        made = [synth.frame_pair(NQ, NT, seed=1234 + 1000 * rank + d, K=K, size=SIZE) for d in range(distinct)]
        sel = [i % distinct for i in range(pairs)]
        up = lambda key: torch.from_numpy(np.stack([made[i][key] for i in sel])).to(dev)
        self.pairs = pairs
        self.q, self.t = up("q"), up("t")                       # (B, N, 128) f32
        self.kp_q, self.kp_t, self.xyz_q = up("kp_q"), up("kp_t"), up("xyz_q")
        nbytes = pipeline.workspace_bytes(NQ, NT, pairs, H, H)
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)


def step(ps: PairSet):
    """One pass over all pairs; everything is enqueued (main stream + the pipeline's PnP side stream),
    nothing is read back.  The PnP tail of one step may still be running when the next step starts; the
    timed region ends with a device-wide synchronize."""
    return pipeline.match_and_ransac(ps.q, ps.t, ps.kp_q, ps.kp_t, RATIO, F_THR, F_CONF, H, xyz_q=ps.xyz_q, K=K,
                                     p_threshold=P_THR, p_confidence=P_CONF, p_iters=H, first_block_pnp=FIRST_BLOCK,
                                     workspace=ps.ws, wait_pnp=False)


def drop_in_latency():
    """Wall clock of the reference-shaped single-pair entry points (host numpy in, host numpy out:
    H2D of the inputs, kernels, stream sync, D2H of the results) on one cfg4-sized frame pair --
    SURVEY.md 8d asks for this beside the device-resident rate; it is not part of `value`."""
    import contextlib, io
    from visual_odometry_amd import cv2_shim
    from visual_odometry_amd.features import match_features_indices
    fp = synth.frame_pair(NQ, NT, seed=77, K=K, size=SIZE)
    out = {}
    def timed(name, fn, reps=3):
        fn()                                                  # warm-up (allocations, first-touch)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name + "_ms"] = float(np.median(ts))
        return r
    with contextlib.redirect_stdout(io.StringIO()):
        qi, ti = timed("match_features_8192x8192", lambda: match_features_indices(fp["q"], fp["t"], RATIO))
        p1, p2 = fp["kp_q"][qi], fp["kp_t"][ti]
        timed("findFundamentalMat_4096_iters", lambda: cv2_shim.findFundamentalMat(p1, p2, cv2_shim.RANSAC, F_THR, F_CONF, H))
        timed("solvePnPRansac_4096_iters", lambda: cv2_shim.solvePnPRansac(fp["xyz_q"][qi], p2, K, None, iterationsCount=H,
                                                                         reprojectionError=P_THR, confidence=P_CONF,
                                                                         flags=cv2_shim.SOLVEPNP_P3P))
    out["matches"] = int(len(qi))
    out["note"] = ("median of 3 calls each, one frame pair, host arrays in and out (PCIe both ways, one stream sync per call); "
                   "the batched device-resident pipeline is what `value` measures")
    return out


def cpu_baseline(max_seconds: float = 25.0):
    """The oracle (CPU restatement of the OpenCV routines; OpenCV itself is absent) on the same
    workload, a bounded number of frame pairs."""
    import oracle

    oracle.build()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    dt, done, parts = 0.0, 0, [0.0, 0.0, 0.0]
    while done < 4 and (done == 0 or dt * (done + 1) / done < max_seconds):
        fp = synth.frame_pair(NQ, NT, seed=1234 + done, K=K, size=SIZE)
        t0 = time.perf_counter()
        idx, dist = oracle.knn2(fp["q"], fp["t"], mt=True)
        keep = oracle.ratio_test(dist, RATIO)
        qi, ti = np.nonzero(keep)[0], idx[keep, 0]
        p1, p2, xyz = fp["kp_q"][qi], fp["kp_t"][ti], fp["xyz_q"][qi]
        t1 = time.perf_counter()
        oracle.find_fundamental_mat(p1, p2, F_THR, F_CONF, H)
        t2 = time.perf_counter()
        oracle.solve_pnp_ransac(xyz, p2, K, H, P_THR, P_CONF)
        t3 = time.perf_counter()
        dt += t3 - t0
        parts = [parts[0] + t1 - t0, parts[1] + t2 - t1, parts[2] + t3 - t2]
        done += 1
    return {"value": done / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame pairs of the same workload (match+gather {parts[0] / done * 1e3:.0f} ms on {threads} "
                      f"OpenMP threads, F-RANSAC {parts[1] / done * 1e3:.0f} ms and PnP-RANSAC incl. EPnP refit "
                      f"{parts[2] / done * 1e3:.0f} ms serial with OpenCV's adaptive early exit); CPU restatement of the "
                      f"OpenCV routines, not OpenCV"}


def iteration_blocks(first):
    out, h0, stage = [], 0, 0
    while h0 < H:
        h1 = H
        if 0 < first < H:
            h1 = first if stage == 0 else min(h0 * 5, H)
        if H - h1 < h1 // 4:
            h1 = H
        out.append((h0, h1)); h0 = h1; stage += 1
    return out


def models_scored(nm, result, first):
    """Models the scoring launches evaluated: samples without a solution have none, and iteration
    blocks that start where OpenCV's loop has already ended are skipped (block k ran iff
    h0_k <= best_iter or h0_k < final niters)."""
    per_pair = []
    for b in range(len(nm)):
        best_iter, niters = int(result[b, 0]), int(result[b, 3])
        per_pair.append(sum(int(nm[b][h0:h1].sum()) for h0, h1 in iteration_blocks(first)
                            if h0 == 0 or h0 <= best_iter or h0 < niters))
    return per_pair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=256, help="frame pairs per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-pair drop_in_latency leg (profiling runs: keeps rocprofv3's per-kernel "
                         "averages to the launches of the steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.require_device(local)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # under torchrun, also with one rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    ps = PairSet(args.pairs, dev, rank)
    lib = _lib.load()
    for _ in range(args.warmup):
        step(ps)
    torch.cuda.synchronize()
    # HIP events around single launches inside the entry points (the scoring kernels of both RANSAC
    # runs and the kNN sweep kernel), recorded on the stream those kernels run on
    cap = 16 * args.steps + 8
    _lib.check(lib.vo_profile_enable((1 << 1) | (1 << 4), cap), "vo_profile_enable")
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(ps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    tags = np.zeros(cap, np.int32); ms = np.zeros(cap, np.float32)
    nrec = lib.vo_profile_collect(tags.ctypes.data, ms.ctypes.data, cap)
    _lib.check(nrec, "vo_profile_collect")
    lib.vo_profile_enable(0, 0)
    # tag 0x1000 = kNN sweep kernel; otherwise kind | iteration-block << 8 of a scoring launch.  The
    # scoring time of one RANSAC run is the sum over its iteration blocks (later blocks return at
    # once for problems that already terminated).
    tot = {"knn2": 0.0, "score_fund": 0.0, "score_pnp": 0.0}
    for i in range(nrec):
        name = "knn2" if tags[i] == 0x1000 else ("score_fund" if (tags[i] & 0xFF) == 0 else "score_pnp")
        tot[name] += float(ms[i]) * 1e-3
    avg = {k: v / args.steps for k, v in tot.items()}

    # ---- what the launches processed, counted outside the timed region from the last step's outputs
    B = args.pairs
    n_match = res.n_match.cpu().numpy().astype(np.int64)
    res_f, res_p = res.f_result.cpu().numpy(), res.p_result.cpu().numpy()
    nm_f, nm_p = [], []
    for b in range(B):
        n = int(n_match[b])
        nm_f.append(ops.solve_7pt(res.f_samples[b], res.pts4[b, :n])[1].cpu().numpy())
        nm_p.append(ops.solve_p3p(res.p_samples[b], res.obj[b, :n], res.img[b, :n], K)[2].cpu().numpy())
    mf, mp = models_scored(nm_f, res_f, 0), models_scored(nm_p, res_p, FIRST_BLOCK)
    evals_f = int(sum(m * n for m, n in zip(mf, n_match)))
    evals_p = int(sum(m * n for m, n in zip(mp, n_match)))
    n_models_f, n_models_p = int(sum(mf)), int(sum(mp))

    prop = torch.cuda.get_device_properties(dev)
    cus = prop.multi_processor_count
    clock_ghz = 2.4
    i8_peak_tops = cus * I8_OPS_PER_CLK_CU * clock_ghz * 1e9 / 1e12

    # algorithmic bytes per (model, correspondence) evaluation = 16 B (x1 y1 x2 y2 f32)
    # or 20 B (PnP), plus per model its 72 / 96 B read and the 4 B count
    sf_gbs = (evals_f * 16 + n_models_f * (72 + 4)) / avg["score_fund"] / 1e9
    sp_gbs = (evals_p * 20 + n_models_p * (96 + 4)) / avg["score_pnp"] / 1e9
    knn_tops = 2.0 * NQ * NT * 128 * B / avg["knn2"] / 1e12

    # physical HBM traffic per launch: PMC counters need their own rocprofv3 passes, so they are
    # measured separately on the same command (scripts/gpu_pmc.sh) and kept under profiles/
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as fh:
            tj = json.load(fh)
        if args.pairs == tj.get("_pairs", 16):
            traffic = {k: v["hbm_bytes_per_launch"] for k, v in tj.items() if isinstance(v, dict)}
    except (OSError, ValueError, KeyError):
        pass
    tnote = "HBM bytes per launch from rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes: profiles/r1_traffic.json"
    roof = {
        "score_fund": {"kernel": "score_kernel<F>", "bound": "hbm", "achieved": sf_gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": sf_gbs / HBM_PEAK_GBS, "traffic": traffic.get("score_fund_kernel"),
                       "traffic_note": tnote,
                       "note": "effective streaming bytes (16 B per model x correspondence); the tiled kernel "
                               "re-uses points from LDS so physical HBM traffic is far lower and the real limiter "
                               "is VALU issue: ~25 f32 ops per evaluation in the filter tier, the f64 filter / exact "
                               "f64 evaluation only within rounding distance of the threshold (DESIGN.md 7)",
                       "f64_tflops_equivalent": evals_f * 49 / avg["score_fund"] / 1e12,
                       "f64_equivalent_note": "49 f64 flops per evaluation of the scalar definition whose verdicts the kernel "
                                              "reproduces bit for bit; not executed as such (f64 vector peak is 78.6 TFLOP/s)",
                       "units_per_launch": f"{evals_f} evaluations = {n_models_f} models x the pair's correspondences ({B} frame pairs)"},
        "score_pnp": {"kernel": "score_kernel<PnP>", "bound": "hbm", "achieved": sp_gbs, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": sp_gbs / HBM_PEAK_GBS, "traffic": None,
                      "note": "the PnP chain runs on a side stream underneath the F chain's getSubset/solver kernels, "
                              "so these launches share the GPU and their durations are upper bounds",
                      "units_per_launch": f"{evals_p} evaluations = {n_models_p} models x the pair's correspondences "
                                          f"({B} frame pairs, {len(iteration_blocks(FIRST_BLOCK))} iteration-block launches summed)"},
        "knn2": {"kernel": "knn2_partial_kernel", "bound": "mfma", "achieved": knn_tops, "peak": i8_peak_tops,
                 "unit": "TOP/s", "frac": knn_tops / i8_peak_tops, "traffic": traffic.get("knn2_partial_kernel"),
                 "note": f"int8 dense peak = {cus} CU x {I8_OPS_PER_CLK_CU} ops/clk x {clock_ghz} GHz; the sweep is bound by "
                         "VALU issue (2.5 integer ops per distance for the exact top-2), see DESIGN.md",
                 "units_per_launch": f"{B} frame pairs of {NQ} x {NT} descriptors"},
    }
    dominant = max(tot, key=tot.get)
    pairs_total = B * world * args.steps
    line = {
        "metric": "matched frame-pairs/sec", "value": pairs_total / dt, "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8 (matching) / f64 (RANSAC; F scoring filtered in f32, verdicts identical)",
        "data": "synthetic",
        "config": {"workload": "cfg4: 8192x8192x128 SIFT match + ratio test -> matched correspondences (on device) -> "
                               "4096-iteration 7-point F-RANSAC and 4096-iteration P3P PnP-RANSAC (+ EPnP refit of its inliers) over them, per frame pair",
                   "pairs_per_gpu_per_step": B, "parallelism": f"pairs sharded over {world} GPU(s), no collective"},
        "ransac_scoring": {"metric": "RANSAC hypotheses x points scored/sec",
                           "value": evals_f / avg["score_fund"],
                           "unit": "evaluations/s per GPU inside the F scoring kernel (models that exist x correspondences)",
                           "pnp_value": evals_p / avg["score_pnp"],
                           "whole_job_evals_per_s": (evals_f + evals_p) * world * args.steps / dt,
                           "matches_per_pair": float(n_match.mean()),
                           "models_scored_per_pair": {"fund": n_models_f / B, "pnp": n_models_p / B},
                           "inliers_per_pair": {"fund": float(res_f[:, 2].mean()), "pnp": float(res_p[:, 2].mean())},
                           "final_niters_first_pairs": {"fund": [int(v) for v in res_f[:4, 3]], "pnp": [int(v) for v in res_p[:4, 3]]}},
        "kernel_time_share": {k: v / dt for k, v in tot.items()},
        "kernel_avg_us": {k: v * 1e6 for k, v in avg.items()},
        "roofline": roof[dominant],
        "roofline_other": [roof[k] for k in roof if k != dominant],
        "device": prop.name, "cus": cus,
    }
    if rank == 0:
        line["drop_in_latency"] = drop_in_latency() if (world == 1 and not args.no_latency) else None
        line["cpu_baseline"] = cpu_baseline() if (world == 1 and not args.no_cpu) else None
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
