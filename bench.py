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
P_FIRST_BLOCK = 256          # PnP iterations evaluated before the first early-termination check

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
        up = lambda lst: torch.from_numpy(np.stack([lst[i] for i in sel])).to(dev)
        self.pairs = pairs
        self.q, self.t = up(qs), up(ts)                         # (B, N, 128) f32
        self.f_pts, self.f_sub = up(f_pts), up(f_sub)           # (B, NPTS, 4) f32, (B, H, 7) i32
        self.p_obj, self.p_img, self.p_sub = up(p_obj), up(p_img), up(p_sub)
        lib = _lib.load()
        nbytes = max(16, lib.vo_knn2_workspace_bytes(NQ, NT, pairs), lib.vo_ransac_workspace_bytes(0, H, pairs),
                     lib.vo_ransac_workspace_bytes(2, H, pairs))
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)


def step(ps: PairSet):
    """One pass over all pairs: 3 matching launches + 2 x (solve, score, replay, mask), no host sync."""
    B = ps.pairs
    qi8, qn, _ = ops.desc_to_i8(ps.q.reshape(B * NQ, 128))
    ti8, tn, _ = ops.desc_to_i8(ps.t.reshape(B * NT, 128))
    idx, d2 = ops.knn2(qi8.reshape(B, NQ, 128), qn.reshape(B, NQ), ti8.reshape(B, NT, 128), tn.reshape(B, NT), ps.ws)
    keep, nkeep = ops.ratio_test(d2.reshape(B * NQ, 2), 0.8)
    f = ops.ransac_batch(ops.KIND_FUND, ps.f_sub, ps.f_pts, F_THR, F_CONF, H, workspace=ps.ws)
    p = ops.ransac_batch(ops.KIND_PNP, ps.p_sub, ps.p_obj, P_THR, P_CONF, H, img=ps.p_img, K=synth.K_MALAGA, workspace=ps.ws,
                         first_block=P_FIRST_BLOCK)
    return idx, keep, nkeep, f, p


def cpu_baseline(max_seconds: float = 25.0):
    """The oracle (CPU restatement of the OpenCV routines; OpenCV itself is absent) on the same
    workload, a bounded number of frame pairs."""
    import oracle

    oracle.build()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    dt, done, parts = 0.0, 0, [0.0, 0.0, 0.0]
    while done < 4 and (done == 0 or dt * (done + 1) / done < max_seconds):
        q, t = synth.sift_like_pair(NQ, NT, seed=1234 + done)
        sc = synth.two_view_scene(NPTS, K=synth.K_KITTI, size=synth.IMAGE_SIZE["kitti"], seed=4321 + done)
        pp = synth.pnp_scene(NPTS, K=synth.K_MALAGA, size=synth.IMAGE_SIZE["malaga"], seed=8765 + done)
        t0 = time.perf_counter()
        idx, dist = oracle.knn2(q, t, mt=True)
        oracle.ratio_test(dist, 0.8)
        t1 = time.perf_counter()
        oracle.find_fundamental_mat(sc["p1"], sc["p2"], F_THR, F_CONF, H)
        t2 = time.perf_counter()
        oracle.solve_pnp_ransac(pp["obj"], pp["img"], pp["K"], H, P_THR, P_CONF)
        t3 = time.perf_counter()
        dt += t3 - t0
        parts = [parts[0] + t1 - t0, parts[1] + t2 - t1, parts[2] + t3 - t2]
        done += 1
    return {"value": done / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame pairs of the same workload (match {parts[0] / done * 1e3:.0f} ms on {threads} OpenMP "
                      f"threads, F-RANSAC {parts[1] / done * 1e3:.0f} ms and PnP-RANSAC {parts[2] / done * 1e3:.0f} ms "
                      f"serial with OpenCV's adaptive early exit); CPU restatement of the OpenCV routines, not OpenCV"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=64, help="frame pairs per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
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
    _lib.check(lib.vo_profile_enable((1 << 1) | (1 << 4), 16 * args.steps + 8), "vo_profile_enable")
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

    cap = 16 * args.steps + 8
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

    # how many models the scoring launches really evaluated, counted outside the timed region:
    # samples with no real root / no P3P solution have no model, and iteration blocks that start at
    # or after the point where OpenCV's loop ends are skipped (block k ran iff h0_k <= best_iter or
    # h0_k < final niters)
    B = args.pairs

    def blocks(first):
        out, h0, stage = [], 0, 0
        while h0 < H:
            h1 = H
            if 0 < first < H:
                h1 = first if stage == 0 else min(h0 * 5, H)
            if H - h1 < h1 // 4:
                h1 = H
            out.append((h0, h1)); h0 = h1; stage += 1
        return out

    def scored(nm, result, first):
        n = 0
        for b in range(B):
            best_iter, niters = int(result[b, 0]), int(result[b, 3])
            for h0, h1 in blocks(first):
                if h0 == 0 or h0 <= best_iter or h0 < niters:
                    n += int(nm[b][h0:h1].sum())
        return n

    res_f, res_p = res[3][0].cpu().numpy(), res[4][0].cpu().numpy()
    nm_f = [ops.solve_7pt(ps.f_sub[b], ps.f_pts[b])[1].cpu().numpy() for b in range(B)]
    nm_p = [ops.solve_p3p(ps.p_sub[b], ps.p_obj[b], ps.p_img[b], synth.K_MALAGA)[2].cpu().numpy() for b in range(B)]
    n_models_f, n_models_p = scored(nm_f, res_f, 0), scored(nm_p, res_p, P_FIRST_BLOCK)
    good_f = int((res_f[:, 2] > 0).sum()); good_p = int((res_p[:, 2] > 0).sum())

    prop = torch.cuda.get_device_properties(dev)
    cus = prop.multi_processor_count
    clock_ghz = 2.4
    i8_peak_tops = cus * I8_OPS_PER_CLK_CU * clock_ghz * 1e9 / 1e12

    # one scoring launch = all models of the B pairs x NPTS correspondences each; algorithmic bytes
    # per (model, correspondence) evaluation = 16 B (x1 y1 x2 y2 f32) or 20 B (PnP),
    # plus per model its 72 / 96 B read and the 4 B count
    sf_bytes = n_models_f * NPTS * 16 + n_models_f * (72 + 4)
    sf_gbs = sf_bytes / avg["score_fund"] / 1e9
    sp_bytes = n_models_p * NPTS * 20 + n_models_p * (96 + 4)
    sp_gbs = sp_bytes / avg["score_pnp"] / 1e9
    knn_ops = 2.0 * NQ * NT * 128 * args.pairs
    knn_tops = knn_ops / avg["knn2"] / 1e12

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
    dominant = max(tot, key=tot.get)
    roof = {
        "score_fund": {"kernel": "score_fund_kernel", "bound": "hbm", "achieved": sf_gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": sf_gbs / HBM_PEAK_GBS, "traffic": traffic.get("score_fund_kernel"),
                       "traffic_note": "HBM bytes per launch from rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes: profiles/r1_traffic.json",
                       "note": "effective streaming bytes (16 B per model x correspondence); the tiled kernel "
                               "re-uses points from LDS so physical HBM traffic is far lower and the real limiter "
                               "is the f64 VALU",
                       "f64_tflops": n_models_f * NPTS * 49 / avg["score_fund"] / 1e12, "f64_peak_tflops": FP64_PEAK_TFLOPS,
                       "units_per_launch": f"{n_models_f} models x {NPTS} correspondences ({B} frame pairs)"},
        "score_pnp": {"kernel": "score_pnp_kernel", "bound": "hbm", "achieved": sp_gbs, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": sp_gbs / HBM_PEAK_GBS, "traffic": traffic.get("score_pnp_kernel"),
                      "units_per_launch": f"{n_models_p} models x {NPTS} correspondences ({B} frame pairs)"},
        "knn2": {"kernel": "knn2_partial_kernel", "bound": "mfma", "achieved": knn_tops, "peak": i8_peak_tops,
                 "unit": "TOP/s", "frac": knn_tops / i8_peak_tops, "traffic": traffic.get("knn2_partial_kernel"),
                 "note": f"int8 dense peak = {cus} CU x {I8_OPS_PER_CLK_CU} ops/clk x {clock_ghz} GHz",
                 "units_per_launch": f"{B} frame pairs of {NQ} x {NT} descriptors"},
    }
    pairs_total = args.pairs * world * args.steps
    evals_per_step = (n_models_f + n_models_p) * NPTS
    line = {
        "metric": "matched frame-pairs/sec", "value": pairs_total / dt, "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8 (matching) / f64 (RANSAC)",
        "data": "synthetic",
        "config": {"workload": "cfg4: 8192x8192x128 SIFT match + ratio test, then 4096-iteration 7-point F-RANSAC "
                               "(up to 3 models per sample x 4096 pts) and 4096-iteration P3P PnP-RANSAC (4096 x 4096 pts) per frame pair",
                   "pairs_per_gpu_per_step": args.pairs, "parallelism": f"pairs sharded over {world} GPU(s), no collective"},
        "ransac_scoring": {"metric": "RANSAC hypotheses x points scored/sec",
                           "value": n_models_f * NPTS / avg["score_fund"],
                           "unit": "evals/s per GPU inside the F scoring kernel (models that exist x correspondences)",
                           "pnp_value": n_models_p * NPTS / avg["score_pnp"],
                           "models_scored_per_pair": {"fund": n_models_f / B, "pnp": n_models_p / B},
                           "final_niters": {"fund": [int(v) for v in res_f[:, 3]], "pnp": [int(v) for v in res_p[:, 3]]},
                           "pairs_with_a_model": {"fund": good_f, "pnp": good_p},
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
