#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X, frame pairs end to end on the device.

Workloads (`--config`):
  cfg4 (default; BASELINE.json configs[3]): per frame pair, 8192 x 8192 SIFT-like descriptors are matched
       (f32 -> int8 conversion, exact 2-NN on int8 MFMA, ratio test 0.8), the surviving matches are compacted
       into correspondences (about 4.9k per pair), and two 4096-iteration RANSAC runs are evaluated over them:
       7-point fundamental matrix (threshold 0.1 px, confidence 0.9999: the reference's F parameters; up to 3
       models per sample, symmetric epipolar error) and P3P PnP (1 px, 0.9999: the reference's Malaga
       parameters; reprojection error) + the EPnP refit of its inliers.
  cfg5 (BASELINE.json configs[4], one GPU's share): 32768 x 32768 descriptors per pair, 65 536-iteration P3P
       PnP-RANSAC over the matches with 55 % of the landmarks replaced by unrelated points (so OpenCV's
       adaptive loop runs for hundreds of iterations instead of a handful), plus the 4096-iteration F-RANSAC.
Each RANSAC run = sample stream of cv::RNG/getSubset generated on the device for the pair's match count ->
minimal solves -> scoring of every model against every correspondence -> replay of OpenCV's sequential
best-model bookkeeping incl. its adaptive iteration count -> inlier mask of the winner.  One "step" = one
pass over `--pairs` frame pairs with no host synchronisation inside.  Inputs are synthetic, seeded, and
resident in HBM before the timed region.

Multi-GPU (`--gpus N`, one process per GPU, torch.distributed backend nccl = RCCL):
  --mode weak  (default): frame pairs are independent, every rank processes its own `--pairs` pairs, no
               collective on the data path; value = pairs of all ranks / max-over-ranks time.
  --mode shard (BASELINE cfg4 "sharded over 2/4/8 GPUs"): ONE batch of `--pairs` pairs is spread over the
               ranks -- query rows split for the match (one all_gather of the 2-NN rows), the iteration
               range of every F-RANSAC problem split over the ranks (one all_gather of the count tables and one
               all-reduce of the winners, replicated replay), whole PnP problems dealt to the ranks (one
               all_gather of the packed results); strong scaling.

Launch: under torchrun (RANK/LOCAL_RANK/WORLD_SIZE in the environment) this process IS a rank.  Run plainly
(`python bench.py --gpus N`) it is a launcher: it touches no GPU, starts N fresh rank processes of itself
and relays rank 0's line.  `--inproc` runs the single rank in this very process (profilers want the program
itself after `--`).  Every form initialises a real RCCL process group, also with one rank.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RATIO = 0.8                   # reference src/params.py:10
F_THR, F_CONF = 0.1, 0.9999   # reference src/params.py:24-27
P_THR, P_CONF = 1.0, 0.9999   # reference src/params.py:72-75
FIRST_BLOCK = 64              # PnP iterations evaluated before the first early-termination check (then x5 per block)
# F iterations counted exactly before the first replay; the blocks behind it (x16 per block) count only the hypotheses
# whose upper bound beats the best count so far (vo_ransac_batch; 0 = one block, every hypothesis counted exactly)
F_FIRST_BLOCK = int(os.environ.get("VO_BENCH_F_FIRST_BLOCK", "256"))

CONFIGS = {
    "cfg4": dict(nq=8192, nt=8192, f_iters=4096, p_iters=4096, pairs=256, landmark_outliers=0.0),
    "cfg5": dict(nq=32768, nt=32768, f_iters=4096, p_iters=65536, pairs=32, landmark_outliers=0.55),
}

PROFILE_TAG = "r3"            # profiles/<tag>_counters.json, <tag>_score_*_isa.json: what the roofline objects cite
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s HBM3E
I8_OPS_PER_CLK_CU = 8192      # 4 SIMD x 2048 int8 ops/clk (v_mfma_i32_32x32x32_i8: 65536 ops / 32 clk)
VALU_LANES_PER_CLK_CU = 128   # 4 SIMD-32: a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)
FP64_FLOP_PER_CLK_CU = 128    # 78.6 TFLOP/s f64 vector at 256 CU x 2.4 GHz
F16_FLOP_PER_CLK_CU = 4096    # 4 SIMD x 1024 flop/clk (v_mfma_f32_32x32x16_f16: 32768 flop / 32 clk): 2.52 PFLOP/s dense
CLOCK_GHZ = 2.4               # nominal peak clock; the chip holds less under load (profiles/r2_counters.json)


# ----------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs this process may use, found WITHOUT loading torch or any HIP/HSA library (the launcher must stay
    provably GPU-free until its children have been started): the runtime's visibility lists if set, else the KFD
    topology in sysfs (nodes with SIMDs are GPUs).  None = cannot tell (the ranks then fail by themselves)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except (OSError, ValueError):
        return None


def launcher(args, argv) -> int:
    """Start args.gpus rank processes of this script (fresh interpreters: this process loads neither torch nor
    HIP, so nothing GPU-initialised is forked or exec'd), watch them, and if one fails end the others instead of
    leaving them blocked in the rendezvous or a collective."""
    n = args.gpus
    have = visible_gpus()
    if n < 1 or (have is not None and n > max(have, 1)):
        print(f"bench.py: --gpus {n} but {have} GPU(s) visible", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                print(f"bench.py: rank process {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()
                deadline = time.time() + 10
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


# ----------------------------------------------------------------------------------------- worker
class PairSet:
    """Device-resident inputs of `pairs` DISTINCT frame pairs: the first four from the host generator (synth.frame_pair,
    the pairs the CPU baseline runs), the rest from the device generator of the same recipe (sweep.frame_pairs_device)."""

    def __init__(self, cfg, pairs: int, dev, seed_base: int, host_pairs: int = 4):
        import numpy as np
        import torch
        from visual_odometry_amd import pipeline, sweep, synth
        host_pairs = min(host_pairs, pairs)
        K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
        made = [synth.frame_pair(cfg["nq"], cfg["nt"], seed=1234 + seed_base + d, K=K, size=size,
                                 landmark_outlier_frac=cfg["landmark_outliers"]) for d in range(host_pairs)]
        keys = ("q", "t", "kp_q", "kp_t", "xyz_q")
        parts = {k: [torch.from_numpy(np.stack([m[k] for m in made])).to(dev)] for k in keys}
        for lo in range(host_pairs, pairs, 32):
            ids = [100000 + seed_base * 1000 + i for i in range(lo, min(pairs, lo + 32))]
            g = sweep.frame_pairs_device(ids, cfg["nq"], cfg["nt"], K, size, dev, landmark_outlier_frac=cfg["landmark_outliers"])
            for k in keys:
                parts[k].append(g[k])
        self.pairs, self.K = pairs, K
        self.q, self.t = torch.cat(parts["q"]), torch.cat(parts["t"])         # (B, N, 128) f32
        self.kp_q, self.kp_t, self.xyz_q = torch.cat(parts["kp_q"]), torch.cat(parts["kp_t"]), torch.cat(parts["xyz_q"])
        nbytes = pipeline.workspace_bytes(cfg["nq"], cfg["nt"], pairs, cfg["f_iters"], cfg["p_iters"])
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.prepared = [None, None]                            # descriptor prefetch buffers (step(prefetch=True))


E_THR, E_CONF, E_MAX_ITERS = 1.0, 0.9999, 50000     # reference src/utils/geometry.py:43 defaults (1 px, 0.9999, 50000 iterations)
E_SAMPLES = 4096              # getSubset samples evaluated per pair for the E chain (OpenCV's adaptive loop must end within them)


def step(ps: PairSet, cfg, group, prefetch: bool = False, pose_tail: bool = True, ess: bool = False):
    """One pass over all pairs; everything is enqueued, nothing is read back.  Weak mode: main stream + the
    pipeline's PnP side stream (the PnP tail of one step may still be running when the next step starts; the
    timed region ends with a device-wide synchronize).  Shard mode: one stream, collectives inside.
    prefetch (weak mode): the way a driver walks through many batches -- the HBM-bound descriptor conversion of the
    NEXT step's batch is enqueued on the prefetch stream (two buffers, used alternately) and runs underneath this
    step's VALU-bound kernels; every step still converts one batch and consumes one conversion."""
    from visual_odometry_amd import pipeline
    prepared = None
    if prefetch:
        if ps.prepared[0] is None:                          # very first step: nothing was prefetched for it
            ps.prepared[0] = pipeline.prepare_descriptors(ps.q, ps.t)
        prepared = ps.prepared[0]
    if ess:
        # the essential-matrix form of the 2D-2D chain (reference src/utils/geometry.py:43-45, dormant there): five-point
        # E-RANSAC with the reference's defaults instead of the 7-point F-RANSAC; the chain ends at (E, mask) as the
        # reference's wrapper does
        return pipeline.match_and_ransac(ps.q, ps.t, ps.kp_q, ps.kp_t, RATIO, F_THR, F_CONF, 0, xyz_q=ps.xyz_q,
                                         K=ps.K, p_threshold=P_THR, p_confidence=P_CONF, p_iters=cfg["p_iters"],
                                         first_block_pnp=FIRST_BLOCK, workspace=ps.ws, wait_pnp=False, prepared=prepared,
                                         e_iters=E_SAMPLES, e_threshold=E_THR, e_confidence=E_CONF, e_max_iters=E_MAX_ITERS,
                                         first_block_e=FIRST_BLOCK)
    res = pipeline.match_and_ransac(ps.q, ps.t, ps.kp_q, ps.kp_t, RATIO, F_THR, F_CONF, cfg["f_iters"], xyz_q=ps.xyz_q,
                                    K=ps.K, p_threshold=P_THR, p_confidence=P_CONF, p_iters=cfg["p_iters"],
                                    first_block_pnp=FIRST_BLOCK, first_block_f=F_FIRST_BLOCK if group is None else 0,
                                    workspace=ps.ws, wait_pnp=False, group=group, prepared=prepared)
    if pose_tail:
        # "R,t out" of the 2D-2D chain (reference src/twoDtwoD.py:57-79): E = K^T F K -> decomposeEssentialMatrix ->
        # disambiguateRelativePose over the F inliers, on the device, behind the F-RANSAC.  Weak mode: on a side stream
        # that the step does not wait for (like the PnP chain: 9.40 vs 9.55 ms per step with it on the caller's stream);
        # the timed region ends with a device-wide synchronize
        res.tail = pipeline.pose_tail(res, ps.K, wait=group is not None)
    if prefetch:
        ps.prepared[1] = pipeline.prepare_descriptors(ps.q, ps.t, reuse=ps.prepared[1])     # for the next step
        ps.prepared.reverse()
    return res


def drop_in_latency(cfg):
    """Wall clock of the reference-shaped single-pair entry points (host numpy in, host numpy out:
    H2D of the inputs, kernels, stream sync, D2H of the results) on one cfg4-sized frame pair --
    SURVEY.md 8d asks for this beside the device-resident rate; it is not part of `value`."""
    import contextlib
    import io
    import numpy as np
    import torch
    from visual_odometry_amd import cv2_shim, synth
    from visual_odometry_amd.features import match_features_indices
    K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
    fp = synth.frame_pair(cfg["nq"], cfg["nt"], seed=77, K=K, size=size)
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
    H = cfg["f_iters"]
    with contextlib.redirect_stdout(io.StringIO()):
        qi, ti = timed(f"match_features_{cfg['nq']}x{cfg['nt']}", lambda: match_features_indices(fp["q"], fp["t"], RATIO))
        p1, p2 = fp["kp_q"][qi], fp["kp_t"][ti]
        timed(f"findFundamentalMat_{H}_iters", lambda: cv2_shim.findFundamentalMat(p1, p2, cv2_shim.RANSAC, F_THR, F_CONF, H))
        timed(f"solvePnPRansac_{cfg['p_iters']}_iters",
              lambda: cv2_shim.solvePnPRansac(fp["xyz_q"][qi], p2, K, None, iterationsCount=cfg["p_iters"],
                                              reprojectionError=P_THR, confidence=P_CONF, flags=cv2_shim.SOLVEPNP_P3P))
    out["matches"] = int(len(qi))
    out["note"] = ("median of 3 calls each, one frame pair, host arrays in and out (PCIe both ways, one stream sync per call); "
                   "the batched device-resident pipeline is what `value` measures")
    return out


def cpu_baseline(cfg, max_seconds: float = 25.0, ess: bool = False):
    """The oracle (CPU restatement of the OpenCV routines; OpenCV itself is absent) on the same
    workload, a bounded number of frame pairs."""
    import numpy as np
    import oracle
    from visual_odometry_amd import synth

    oracle.build()
    K, size = synth.K_MALAGA, synth.IMAGE_SIZE["malaga"]
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    dt, done, parts = 0.0, 0, [0.0, 0.0, 0.0]
    while done < 4 and (done == 0 or dt * (done + 1) / done < max_seconds):
        fp = synth.frame_pair(cfg["nq"], cfg["nt"], seed=1234 + done, K=K, size=size,
                              landmark_outlier_frac=cfg["landmark_outliers"])
        t0 = time.perf_counter()
        idx, dist = oracle.knn2(fp["q"], fp["t"], mt=True)
        keep = oracle.ratio_test(dist, RATIO)
        qi, ti = np.nonzero(keep)[0], idx[keep, 0]
        p1, p2, xyz = fp["kp_q"][qi], fp["kp_t"][ti], fp["xyz_q"][qi]
        t1 = time.perf_counter()
        if ess:
            oracle.find_essential_mat(p1.astype(np.float64), p2.astype(np.float64), K, E_CONF, E_THR, E_MAX_ITERS)
        else:
            oracle.find_fundamental_mat(p1, p2, F_THR, F_CONF, cfg["f_iters"])
        t2 = time.perf_counter()
        oracle.solve_pnp_ransac(xyz, p2, K, cfg["p_iters"], P_THR, P_CONF)
        t3 = time.perf_counter()
        dt += t3 - t0
        parts = [parts[0] + t1 - t0, parts[1] + t2 - t1, parts[2] + t3 - t2]
        done += 1
    return {"value": done / dt, "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame pairs of the same workload (match+gather {parts[0] / done * 1e3:.0f} ms on {threads} "
                      f"OpenMP threads, {'E' if ess else 'F'}-RANSAC {parts[1] / done * 1e3:.0f} ms and PnP-RANSAC incl. EPnP refit "
                      f"{parts[2] / done * 1e3:.0f} ms serial with OpenCV's adaptive early exit); CPU restatement of the "
                      f"OpenCV routines, not OpenCV"}


def iteration_blocks(H, first, grow=5):
    out, h0, stage = [], 0, 0
    while h0 < H:
        h1 = H
        if 0 < first < H:
            h1 = first if stage == 0 else min(h0 * grow, H)
        if H - h1 < h1 // 4:
            h1 = H
        out.append((h0, h1)); h0 = h1; stage += 1
    return out


def models_scored(nm, result, H, first, grow=5):
    """Models the scoring launches evaluated: samples without a solution have none, and iteration
    blocks that start where OpenCV's loop has already ended are skipped (block k ran iff
    h0_k <= best_iter or h0_k < final niters)."""
    per_pair = []
    for b in range(len(nm)):
        best_iter, niters = int(result[b, 0]), int(result[b, 3])
        per_pair.append(sum(int(nm[b][h0:h1].sum()) for h0, h1 in iteration_blocks(H, first, grow)
                            if h0 == 0 or h0 <= best_iter or h0 < niters))
    return per_pair


def _load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def worker(args) -> int:
    import numpy as np
    import torch
    import torch.distributed as dist
    from visual_odometry_amd import _lib, ops, parallel

    cfg = dict(CONFIGS[args.config])
    if args.no_pnp:
        cfg["p_iters"] = 0                                    # A/B only: prices what the PnP side stream costs the F chain
    B = args.pairs or cfg["pairs"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    world = int(os.environ.setdefault("WORLD_SIZE", "1"))
    rank = int(os.environ.setdefault("RANK", "0"))
    local = int(os.environ.setdefault("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.require_device(local)
    import datetime
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))   # RCCL, also with one rank
    shard = args.mode == "shard"
    ess = args.ransac == "ess"
    if ess and shard:
        print("bench.py: --ransac ess runs in weak mode only", file=sys.stderr)
        return 2
    group = parallel.WORLD if shard else None
    prefetch = not shard and args.prefetch

    # shard mode: every rank holds the SAME batch (seed_base 0); weak mode: every rank its own pairs
    ps = PairSet(cfg, B, dev, 0 if shard else 1000 * rank)
    lib = _lib.load()
    for _ in range(args.warmup):
        step(ps, cfg, group, prefetch, not args.no_pose_tail, ess)
    torch.cuda.synchronize()
    # HIP events around single launches inside the entry points (the scoring kernels of both RANSAC
    # runs and the kNN sweep kernel), recorded on the stream those kernels run on
    cap = 16 * args.steps + 8
    _lib.check(lib.vo_profile_enable((1 << 1) | (1 << 4), cap), "vo_profile_enable")
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(ps, cfg, group, prefetch, not args.no_pose_tail, ess)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    tags = np.zeros(cap, np.int32); ms = np.zeros(cap, np.float32)
    nrec = lib.vo_profile_collect(tags.ctypes.data, ms.ctypes.data, cap)
    _lib.check(nrec, "vo_profile_collect")
    lib.vo_profile_enable(0, 0)
    res.check_descriptors()                                   # the batched path only flags bad descriptors: raise here
    # tag 0x1000 = kNN sweep kernel; otherwise kind | iteration-block << 8 of a scoring launch.  The
    # scoring time of one RANSAC run is the sum over its iteration blocks (later blocks return at
    # once for problems that already terminated).
    tv = "ess" if ess else "fund"                              # the two-view chain of this run
    tot = {"knn2": 0.0, "score_" + tv: 0.0, "score_pnp": 0.0}
    for i in range(nrec):
        name = "knn2" if tags[i] == 0x1000 else ("score_pnp" if (tags[i] & 0xFF) == 2 else "score_" + tv)
        tot[name] += float(ms[i]) * 1e-3
    avg = {k: v / args.steps for k, v in tot.items()}

    # ---- what this rank's launches processed, counted outside the timed region from the last step's outputs
    NQ, NT, HP = cfg["nq"], cfg["nt"], cfg["p_iters"]
    HF = E_SAMPLES if ess else cfg["f_iters"]                  # samples of the two-view chain
    n_match = res.n_match.cpu().numpy().astype(np.int64)
    res_f = (res.e_result if ess else res.f_result).cpu().numpy()
    res_p = res.p_result.cpu().numpy() if res.p_result is not None else np.zeros((B, 4), np.int32)
    # shard mode: this rank scores its slice of every F problem's iterations and whole PnP problems of its pairs
    h_f = parallel.shard_bounds(HF, rank, world) if shard else (0, HF)
    p_pairs = range(*parallel.shard_bounds(B, rank, world)) if shard else range(B)
    # one solver launch per chain over all pairs (pair b's sample indices shifted into the flattened point array), so
    # that the kernel trace of a bench run holds no per-pair accounting launches
    def count_models(solve, samples, *arrays):
        nb, hh, m = samples.shape
        stride = arrays[0].shape[1]
        shift = (torch.arange(nb, device=samples.device, dtype=torch.int32) * stride).view(nb, 1, 1)
        out = solve((samples + shift).reshape(nb * hh, m), *[a.reshape(nb * stride, -1) for a in arrays])
        return list(out[-1].reshape(nb, hh).cpu().numpy())
    if ess:
        nm_f = count_models(ops.solve_5pt, res.e_samples, res.pts4n)
    else:
        nm_f = count_models(ops.solve_7pt, res.f_samples[:, h_f[0]:h_f[1]], res.pts4)
    nm_p = []
    if HP > 0:
        sel = torch.tensor(list(p_pairs), dtype=torch.long, device=dev)
        nm_p = count_models(lambda sm, o, i: ops.solve_p3p(sm, o, i, ps.K), res.p_samples[sel], res.obj[sel], res.img[sel])
    mf = [int(x.sum()) for x in nm_f] if shard else (models_scored(nm_f, res_f, HF, FIRST_BLOCK) if ess else models_scored(nm_f, res_f, HF, F_FIRST_BLOCK, grow=16))
    mp = models_scored(nm_p, res_p[list(p_pairs)], HP, FIRST_BLOCK) if HP > 0 else []
    evals_f = int(sum(m * n for m, n in zip(mf, n_match)))
    evals_p = int(sum(m * n for m, n in zip(mp, n_match[list(p_pairs)])))
    n_models_f, n_models_p = int(sum(mf)), int(sum(mp))
    nq_local = (parallel.shard_bounds(NQ, rank, world)[1] - parallel.shard_bounds(NQ, rank, world)[0]) if shard else NQ
    unfinished = int((res_f[:, 3] > HF).sum()) if ess else 0   # E pairs whose adaptive loop wants more samples than were evaluated

    prop = torch.cuda.get_device_properties(dev)
    cus = prop.multi_processor_count
    i8_peak_tops = cus * I8_OPS_PER_CLK_CU * CLOCK_GHZ * 1e9 / 1e12
    valu_peak = cus * VALU_LANES_PER_CLK_CU * CLOCK_GHZ * 1e9 / 1e12        # T lane-instructions/s
    fp64_peak = cus * FP64_FLOP_PER_CLK_CU * CLOCK_GHZ * 1e9 / 1e12

    # physical HBM traffic, SQ instruction counts and the clock under load come from rocprofv3 --pmc passes of
    # this same command (scripts/gpu_roofline.sh -> profiles/r3_counters.json); the per-evaluation instruction counts
    # of the scoring loops from the disassembled code object (scripts/isa_count.py -> profiles/r3_score_*_isa.json)
    ctr = _load_json(f"{PROFILE_TAG}_counters.json") or {}
    same_shape = ctr.get("_pairs") == B and ctr.get("_config", "cfg4") == args.config and not shard and not ess
    kc = lambda k: (ctr.get(k) or {}) if same_shape else {}
    t_f = max(avg["score_" + tv], 1e-12)
    t_p = max(avg["score_pnp"], 1e-12)
    t_k = max(avg["knn2"], 1e-12)
    tiers = {"fund": "tier 1a, the one-sided sure-outlier test every evaluation goes through; the ~2.5 % of evaluations that go on to "
                     "the pooled f32 filter and the ~0.2 % that reach the f64 / exact tier are overhead, not counted",
             "ess": "the two-sided f32 sign-bit Sampson filter every evaluation goes through; the exact f64 evaluation of the few it "
                    "cannot decide is overhead, not counted",
             "pnp": "the division-free f32 sign-bit reprojection filter every evaluation goes through; the exact f64 projection of "
                    "the few it cannot decide is overhead, not counted"}

    def valu_roofline(kind, template_arg, evals, n_models, t, stream_bytes, model_bytes, ckey):
        isa = _load_json(f"{PROFILE_TAG}_score_{kind}_isa.json") or {}
        vpe = isa.get("valu_per_evaluation")
        r = {"kernel": f"score_kernel<{template_arg}>", "bound": "valu-issue", "unit": "T VALU lane-instructions/s", "peak": valu_peak,
             "traffic": kc(ckey).get("hbm_bytes_per_launch"),
             "units_per_launch": f"{evals} (model, correspondence) evaluations = {n_models} models x the pair's correspondences ({B} frame pairs)"}
        if vpe and evals:
            r["achieved"] = evals * vpe / t / 1e12
            r["frac"] = r["achieved"] / valu_peak
            r["valu_instructions_per_evaluation"] = vpe
            r["all_instructions_per_evaluation"] = isa.get("all_per_evaluation")
            r["note"] = (f"achieved = evaluations x {vpe} VALU instructions per evaluation ({tiers[kind]}: inner loop of the shipped "
                         f"code object, profiles/{PROFILE_TAG}_score_{kind}_isa.json; one lane = one model) / launch time; peak = {cus} CU x 4 SIMD x "
                         f"32 lanes/clk x {CLOCK_GHZ} GHz (a wave64 VALU instruction issues over 2 cycles).  The kernel keeps 256 models in "
                         "registers per workgroup and re-reads correspondences from LDS, so it is bound by instruction issue, not by HBM "
                         "(traffic = measured HBM bytes per launch)")
            c = kc(ckey)
            if c.get("clock_ghz"):
                r["clock_ghz_under_load"] = c["clock_ghz"]
                r["frac_at_measured_clock"] = r["achieved"] / (cus * VALU_LANES_PER_CLK_CU * c["clock_ghz"] * 1e9 / 1e12)
            if c.get("SQ_INSTS_VALU") and c.get("evaluations_per_launch"):
                r["valu_per_evaluation_from_SQ_INSTS_VALU"] = c["SQ_INSTS_VALU"] * 64 / c["evaluations_per_launch"]
        else:
            r["achieved"], r["frac"] = None, None
            r["note"] = f"profiles/{PROFILE_TAG}_score_{kind}_isa.json missing (run scripts/isa_count.py) or nothing was scored"
        r["effective_streaming_GBps"] = (evals * stream_bytes + n_models * (model_bytes + 4)) / t / 1e9
        r["effective_streaming_note"] = (f"SURVEY.md 8d's figure: {stream_bytes} B per evaluation as if every model streamed every "
                                         f"correspondence from HBM; not a roofline for this kernel ({HBM_PEAK_GBS:.0f} GB/s peak is never touched)")
        return r

    sf = valu_roofline(tv, "E" if ess else "F", evals_f, n_models_f, t_f, 32 if ess else 16, 72, f"score_{tv}_kernel")
    isa_f = _load_json(f"{PROFILE_TAG}_score_fund_isa.json") or {}
    if not ess and isa_f.get("mfma_flop_per_evaluation_useful") and evals_f:
        # the F count kernel's first tier runs on the matrix cores (csrc/vo_score_fmfma.h): priced against the dense f16 MFMA peak
        # with the multiply-adds the split inner products need; the VALU-issue view of the same launch is kept beside it
        f16_peak = cus * F16_FLOP_PER_CLK_CU * CLOCK_GHZ * 1e9 / 1e12
        useful, issued, vpe = isa_f["mfma_flop_per_evaluation_useful"], isa_f["mfma_flop_per_evaluation_issued"], isa_f["valu_per_evaluation"]
        c = kc("score_fund_kernel")
        sf = {"kernel": "score_fund_mfma_kernel", "bound": "mfma", "unit": "TFLOP/s", "peak": f16_peak,
              "achieved": evals_f * useful / t_f / 1e12, "traffic": c.get("hbm_bytes_per_launch"),
              "units_per_launch": sf["units_per_launch"], "mfma_flop_per_evaluation": useful,
              "mfma_flop_per_evaluation_issued": issued, "valu_instructions_per_evaluation": vpe,
              "valu_issue_frac": evals_f * vpe / t_f / 1e12 / valu_peak,
              "effective_streaming_GBps": sf["effective_streaming_GBps"], "effective_streaming_note": sf["effective_streaming_note"]}
        sf["frac"] = sf["achieved"] / f16_peak
        sf["frac_of_issued_flops"] = evals_f * issued / t_f / 1e12 / f16_peak
        sf["note"] = (f"achieved = evaluations x {useful} flop per evaluation (d = x2^T F x1 and den2 = |(F x1)_xy|^2 as inner products of f16 hi/lo "
                      f"split coefficient and feature vectors: 26 + 16 multiply-adds; the three v_mfma_f32_32x32x16_f16 per 1024 evaluations issue "
                      f"{issued}) / launch time; peak = {cus} CU x {F16_FLOP_PER_CLK_CU} flop/clk x {CLOCK_GHZ} GHz (dense f16 MFMA).  The VALU side of the "
                      f"tier is {vpe} instructions per evaluation (profiles/{PROFILE_TAG}_score_fund_isa.json), valu_issue_frac of the 78.6 T "
                      "lane-instructions/s peak.  Neither pipe is the limit any more: ~2.4 % of the evaluations are true inliers of their "
                      "hypothesis and take the complete two-sided f32 filter through a per-wave queue (LDS gathers, latency-bound), see "
                      "DESIGN.md 4.2; the scalar kernel this replaces needed 14.125 VALU instructions per evaluation and 1.7x the time")
        pruned = F_FIRST_BLOCK > 0 and not shard
        if pruned:
            # blocks behind the first: every evaluation still goes through the matrix-core tier (screen_fund_mfma_kernel), the
            # second tier runs only for the hypotheses that can still beat the best count (count_contenders_kernel)
            c = kc("screen_fund_kernel")
            sf["kernel"] = ("screen_fund_mfma_kernel (+ score_fund_mfma_kernel for the first block, screen_finalize_kernel, "
                            "count_contenders_kernel: the launches of one pair batch's F scoring summed)")
            both = [kc(k).get("hbm_bytes_per_launch") for k in ("screen_fund_kernel", "score_fund_kernel", "count_contenders_kernel")]
            sf["traffic"] = sum(both) if all(v is not None for v in both) else None
            sf["note"] += (f".  PRUNED BLOCKS (this run): the first {F_FIRST_BLOCK} iterations are counted as described; behind them the first "
                           "tier alone gives every hypothesis an upper bound of its count, and only the ones whose bound beats the best count "
                           "so far are counted exactly -- OpenCV's loop acts on good > maxGoodCount only, so F, mask and the iteration record "
                           "are unchanged (tests); VO_BENCH_F_FIRST_BLOCK=0 times the unpruned form")
        if c.get("clock_ghz"):
            sf["clock_ghz_under_load"] = c["clock_ghz"]
            sf["frac_at_measured_clock"] = sf["achieved"] / (cus * F16_FLOP_PER_CLK_CU * c["clock_ghz"] * 1e9 / 1e12)
        if not pruned and c.get("SQ_INSTS_VALU") and c.get("evaluations_per_launch"):
            sf["valu_per_evaluation_from_SQ_INSTS_VALU"] = c["SQ_INSTS_VALU"] * 64 / c["evaluations_per_launch"]
    sp = valu_roofline("pnp", "PnP", evals_p, n_models_p, t_p, 20, 96, "score_pnp_kernel")
    sp["note"] = (sp.get("note") or "") + ("; in weak mode the PnP chain runs on a side stream under the two-view chain, so these launches "
                                           "share the GPU and their durations are upper bounds (standalone: profiles/r3_score_standalone.json)")
    sp["units_per_launch"] += f", {len(iteration_blocks(HP, FIRST_BLOCK))} iteration-block launches summed"
    roof = {
        "score_" + tv: sf,
        "score_pnp": sp,
        "knn2": {"kernel": "knn2_partial_kernel", "bound": "mfma", "achieved": 2.0 * nq_local * NT * 128 * B / t_k / 1e12,
                 "peak": i8_peak_tops, "unit": "TOP/s", "traffic": kc("knn2_partial_kernel").get("hbm_bytes_per_launch"),
                 "note": f"int8 dense peak = {cus} CU x {I8_OPS_PER_CLK_CU} ops/clk x {CLOCK_GHZ} GHz; under this kernel the chip holds "
                         "1.8-1.9 GHz (power), see clock_ghz_under_load and DESIGN.md 4.1",
                 "units_per_launch": f"{B} frame pairs of {nq_local} x {NT} descriptors"},
    }
    roof["knn2"]["frac"] = roof["knn2"]["achieved"] / i8_peak_tops
    if kc("knn2_partial_kernel").get("clock_ghz"):
        ck = kc("knn2_partial_kernel")["clock_ghz"]
        roof["knn2"]["clock_ghz_under_load"] = ck
        roof["knn2"]["frac_at_measured_clock"] = roof["knn2"]["achieved"] / (cus * I8_OPS_PER_CLK_CU * ck * 1e9 / 1e12)
    dominant = max(tot, key=tot.get)
    pairs_total = B * args.steps * (1 if shard else world)
    f_note = (f"; behind the first {F_FIRST_BLOCK} of them a hypothesis whose upper bound cannot beat the best inlier count so far is not "
              "counted exactly -- F, mask and iteration record identical") if F_FIRST_BLOCK > 0 and not shard else ""
    wl = {"cfg4": f"cfg4: {NQ}x{NT}x128 SIFT match + ratio test -> matched correspondences (on device) -> {HF}-iteration "
                  f"7-point F-RANSAC (thr 0.1 px: every iteration runs{f_note}) and P3P PnP-RANSAC with maxIters {HP} (+ EPnP refit of "
                  "its inliers) over them, per frame pair; ~91 % of the PnP correspondences are inliers, so OpenCV's adaptive "
                  "loop -- which the pipeline reproduces -- ends after 3-12 iterations: the first block of 64 iterations is what gets scored",
          "cfg5": f"cfg5 (one GPU's share of the sweep): {NQ}x{NT}x128 SIFT match + ratio test -> matched correspondences -> "
                  f"P3P PnP-RANSAC with maxIters {HP} over them with 55 % of the landmarks replaced by unrelated points "
                  f"(OpenCV's adaptive loop runs a few hundred iterations; blocks of 64, 320, 1600, ... are scored) + EPnP refit, "
                  f"and the {HF}-iteration 7-point F-RANSAC, per frame pair"}[args.config]
    if ess:
        wl = (f"{args.config} with the essential-matrix form of the 2D-2D chain (reference src/utils/geometry.py:43-45): {NQ}x{NT}x128 SIFT "
              f"match + ratio test -> matched correspondences -> five-point E-RANSAC (cv2.findEssentialMat defaults of the reference: "
              f"threshold {E_THR} px, confidence {E_CONF}, maxIters {E_MAX_ITERS}; Sampson distance on K-normalised f64 coordinates; "
              f"{E_SAMPLES} getSubset samples per pair in blocks of {FIRST_BLOCK}, x5, ...: OpenCV's adaptive loop ends after a few dozen) "
              f"and P3P PnP-RANSAC with maxIters {HP} (+ EPnP refit), per frame pair; the chain ends at (E, mask) as the reference's wrapper does")
    else:
        wl += ("; both chains end at a pose: the 2D-2D chain goes on through E = K^T F K, decomposeEssentialMatrix and "
               "disambiguateRelativePose over the F inliers (device pose tail on a side stream, R,t + triangulated points out), the "
               "3D-2D chain ends with the EPnP refit" if not args.no_pose_tail else "; the 2D-2D chain stops at F (--no-pose-tail)")
    par_note = (f"ONE batch of {B} pairs over {world} GPU(s): query rows split for the match (all_gather of the 2-NN rows), "
                "F-RANSAC iteration ranges split over the ranks (all_gather of count tables + all-reduce of winners, replicated replay), "
                "PnP problems dealt to the ranks by pair (all_gather of the packed results)"
                if shard else f"pairs sharded over {world} GPU(s), no collective on the data path")
    line = {
        "metric": "matched frame-pairs/sec", "value": pairs_total / dt, "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
        "dtype": "i8 (matching) / f64 (RANSAC; scoring filtered by f16 hi/lo-split MFMA and f32 tiers, verdicts identical)",
        "data": "synthetic",
        "data_note": "every frame pair of a step is a different synthetic scene (4 from the host generator -- the ones the CPU "
                     "baseline runs -- the rest from the device generator of the same recipe); random-init, no dataset",
        "config": {"workload": wl, "pairs_per_step": B if shard else B * world, "pairs_per_gpu_per_step": None if shard else B,
                   "mode": args.mode, "two_view_chain": "essential (5-point, Sampson)" if ess else "fundamental (7-point, symmetric epipolar)",
                   "pose_tail_in_step": not args.no_pose_tail and not ess, "f_first_block": (F_FIRST_BLOCK if not shard and not ess else 0), "descriptor_prefetch": prefetch, "parallelism": par_note, "process_group": f"nccl (RCCL), world size {world}"},
        "ransac_scoring": {"metric": "RANSAC hypotheses x points scored/sec",
                           "value": evals_f / t_f,
                           "unit": f"evaluations/s per GPU inside the {'E' if ess else 'F'} scoring kernel (models that exist x correspondences)",
                           "two_view_pairs_needing_more_samples": unfinished,
                           "pnp_value": evals_p / t_p,
                           "whole_job_evals_per_s": (evals_f + evals_p) * world * args.steps / dt,
                           "matches_per_pair": float(n_match.mean()),
                           "models_scored_per_pair": {tv: n_models_f / B, "pnp": n_models_p / B},
                           "inliers_per_pair": {tv: float(res_f[:, 2].mean()), "pnp": float(res_p[:, 2].mean())},
                           "final_niters_first_pairs": {tv: [int(v) for v in res_f[:4, 3]], "pnp": [int(v) for v in res_p[:4, 3]]}},
        "kernel_time_share": {k: v / dt for k, v in tot.items()},
        "kernel_avg_us": {k: v * 1e6 for k, v in avg.items()},
        "roofline": roof[dominant],
        "roofline_other": [roof[k] for k in roof if k != dominant],
        "device": prop.name, "cus": cus,
    }
    if rank == 0:
        solo = world == 1 and args.config == "cfg4" and not ess
        line["drop_in_latency"] = drop_in_latency(cfg) if (solo and not args.no_latency) else None
        line["cpu_baseline"] = cpu_baseline(cfg, ess=ess) if (world == 1 and not args.no_cpu) else None
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg4")
    ap.add_argument("--mode", choices=["weak", "shard"], default="weak")
    ap.add_argument("--pairs", type=int, default=0, help="frame pairs per GPU per step (weak) / per step (shard); 0 = the config's default")
    ap.add_argument("--inproc", action="store_true", help="be the single rank in this process (for profilers)")
    ap.add_argument("--prefetch", action="store_true",
                    help="weak mode: convert the next step's descriptors underneath this step's scoring (pipeline.prepare_descriptors; "
                         "measured neutral: 9.17 vs 9.16 ms per step, DESIGN.md section 7)")
    ap.add_argument("--ransac", choices=["fund", "ess"], default="fund",
                    help="the two-view chain: 7-point F-RANSAC (the reference's live path, default) or five-point E-RANSAC with the "
                         "reference's calc_essential_mat defaults (its dormant path)")
    ap.add_argument("--no-pose-tail", action="store_true",
                    help="end the 2D-2D chain at F (round-2 workload) instead of at the relative pose (A/B of the pose tail's cost)")
    ap.add_argument("--no-pnp", action="store_true", help="A/B only: drop the PnP chain (side stream) from the step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-pair drop_in_latency leg (profiling runs: keeps rocprofv3's per-kernel "
                         "averages to the launches of the steps)")
    args = ap.parse_args()
    if "RANK" in os.environ or args.inproc:
        return worker(args)
    return launcher(args, sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())
