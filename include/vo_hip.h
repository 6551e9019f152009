/*
 * vo_hip.h -- C ABI of libvo_hip.so, the MI355X (gfx950) implementation of the
 * data-parallel front end of Heusini/visual_odometry.
 *
 * The reference has no FFI of its own: its hot path is three Python calls into OpenCV
 * 4.6.0.  Each entry point below names the reference call site it stands in for
 * (paths relative to the reference repository) and the OpenCV routine behind it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless its name ends in
 *     `_host`; the library never allocates or frees caller-visible memory;
 *   - every device entry point takes a hipStream_t (passed as void*), enqueues work on it
 *     and returns without synchronising; host-only helpers (vo_host_*) touch no GPU state;
 *   - return value: 0 = VO_OK, negative = VO_E* below; vo_last_error() gives the text;
 *   - scratch memory comes in through an explicit workspace argument whose size the
 *     matching *_workspace_bytes() call reports;
 *   - thread-safe for distinct streams.  Global state: the per-thread last-error string; the process-wide
 *     record list of the measurement aid vo_profile_* (off by default, not thread-safe); and, per GPU,
 *     the fixed scratch array of vo_ransac_batch_static (calls on one GPU share it).
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK            0
#define VO_EINVAL       -1   /* bad argument (null pointer, negative size, ...)      */
#define VO_ESIZE        -2   /* size outside what the kernels support                */
#define VO_EWORKSPACE   -3   /* workspace too small                                  */
#define VO_EHIP         -4   /* a HIP runtime call failed                            */
#define VO_ENODEVICE    -5   /* no gfx950 device / kernels cannot load               */

#define VO_DESC_DIM     128  /* SIFT descriptor length (reference src/features.py:33-41) */
#define VO_KNN_MAX_TRAIN 262144 /* BFMatcher's own limit on train rows  */

/* bits of the flag word written by vo_desc_f32_to_i8 */
#define VO_DESC_BAD_VALUE  1   /* an element is not an integer in [0,255]            */
#define VO_DESC_BAD_NORM   2   /* a row has ||desc||_2 > 1024: integer keys could mis-order
                                  where OpenCV's f32 sqrt collides      */

const char* vo_last_error(void);
int  vo_version(void);
/* 0 when a gfx950 device is visible and the code object loads, VO_ENODEVICE otherwise */
int  vo_device_check(int device);

/* ------------------------------------------------------------------------------------
 * Descriptor matching.  Replaces cv.BFMatcher().knnMatch(des_i, des_j, k=2) at reference
 * src/features.py:100-101 (OpenCV features2d BFMatcher::knnMatchImpl -> core batchDistance).
 * ---------------------------------------------------------------------------------- */

/* f32 descriptors holding integers 0..255 (what cv.SIFT produces) -> int8 (value-128) and
 * the squared norms of the int8 rows.  flags (1 x int32, caller zeroes) receives VO_DESC_BAD_*. */
int vo_desc_f32_to_i8(const float* desc, int n, int8_t* out_i8, int32_t* out_norm,
                      int32_t* flags, void* stream);

size_t vo_knn2_workspace_bytes(int nq, int nt, int batch);

/* For each of `batch` independent (query set, train set) pairs, all of the same shape:
 *   q      batch x nq x 128 int8     qnorm  batch x nq int32
 *   t      batch x nt x 128 int8     tnorm  batch x nt int32
 *   idx    batch x nq x 2 int32      train indices of the nearest and second nearest row,
 *                                    ties to the lower index; -1 where nt < 2 leaves a gap
 *   d2     batch x nq x 2 int32      exact squared L2 distances (INT32_MAX where idx = -1)
 * int8 MFMA, int32 accumulation: D = |q|^2 + |t|^2 - 2 q.t exactly. */
int vo_knn2_l2_i8(const int8_t* q, const int32_t* qnorm, int nq,
                  const int8_t* t, const int32_t* tnorm, int nt,
                  int32_t* idx, int32_t* d2, int batch,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Lowe ratio test of reference src/features.py:103-106 on device:
 *   keep[i] = (double)sqrtf(d2[i][0]) < ratio * (double)sqrtf(d2[i][1])   (strict)
 * keep: n x uint8.  count (1 x int32, caller zeroes) receives the number kept. */
int vo_ratio_test(const int32_t* d2, int n, double ratio, uint8_t* keep, int32_t* count,
                  void* stream);

/* ------------------------------------------------------------------------------------
 * RANSAC hypothesis scoring.  Replaces the computeError + findInliers pair that
 * RANSACPointSetRegistrator::run executes once per model (OpenCV calib3d ptsetreg.cpp),
 * i.e. the inside of
 *   cv2.findFundamentalMat(..., cv2.RANSAC, ...)   reference src/utils/geometry.py:39
 *   cv2.findEssentialMat(..., cv2.RANSAC, ...)     reference src/utils/geometry.py:44
 *   cv.solvePnPRansac(..., SOLVEPNP_P3P, ...)      reference src/pnp.py:16-26
 * thr2 is (float)(threshold*threshold); a point is an inlier iff (float)err <= thr2.
 * counts[h] = number of inliers of model h, or -1 when valid[h] == 0 (valid may be NULL).
 * The counts are those of the scalar definitions, bit for bit: the kernels settle most correspondences with
 * division-free f32 filters whose error bounds are rigorous and evaluate the definition itself only where a
 * filter cannot tell (within rounding distance of the threshold; zero, non-finite or absurdly scaled models and
 * coordinates).  The vo_mask_* kernels always evaluate the definitions.
 * ---------------------------------------------------------------------------------- */

/* models H x 9 f64 (row-major F), pts N x 4 f32 (x1 y1 x2 y2).  Symmetric squared
 * point-to-epipolar-line distance (FMEstimatorCallback::computeError, fundam.cpp). */
int vo_score_fund(const double* models, const uint8_t* valid, int H,
                  const float* pts, int N, float thr2, int32_t* counts, void* stream);
/* Upper bounds instead of counts: bound[h] >= the count vo_score_fund would give model h, from the first (matrix-core)
 * tier of the F count kernel alone -- the correspondences it cannot prove to be outliers.  Inside cv2.findFundamentalMat's
 * RANSAC loop (reference src/utils/geometry.py:21-23; RANSACPointSetRegistrator::run acts only on
 * goodCount > maxGoodCount) a hypothesis whose bound does not exceed the best count so far needs no exact count:
 * vo_ransac_batch uses this for the iteration blocks behind the first.  No `valid` array: every model is bounded. */
int vo_score_fund_bound(const double* models, int H, const float* pts, int N, float thr2, int32_t* bound, void* stream);
/* models H x 9 f64 (row-major E), pts N x 4 f64 (K-normalised x1 y1 x2 y2).  Sampson
 * error (EMEstimatorCallback::computeError, five-point.cpp). */
int vo_score_ess(const double* models, const uint8_t* valid, int H,
                 const double* pts, int N, float thr2, int32_t* counts, void* stream);
/* models H x 12 f64 (R row-major, then t), K4_host = {fx, fy, cx, cy} on the HOST,
 * obj N x 3 f32, img N x 2 f32.  Squared reprojection error, f64 projection rounded to f32
 * (PnPRansacCallback::computeError, solvepnp.cpp -> projectPoints, calibration.cpp). */
int vo_score_pnp(const double* models, const uint8_t* valid, int H,
                 const float* obj, const float* img, const double* K4_host, int N,
                 float thr2, int32_t* counts, void* stream);

/* Inlier mask (N x uint8) of ONE model (device pointer to 9 / 9 / 12 f64). */
int vo_mask_fund(const double* model, const float* pts, int N, float thr2, uint8_t* mask, void* stream);
int vo_mask_ess(const double* model, const double* pts, int N, float thr2, uint8_t* mask, void* stream);
int vo_mask_pnp(const double* model, const float* obj, const float* img, const double* K4_host,
                int N, float thr2, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------
 * Minimal solvers, one hypothesis per sample.  Replace the runKernel callbacks of the same
 * three OpenCV calls: run7Point (fundam.cpp), the five-point solver (five-point.cpp),
 * p3p::solve + ranking by reprojection error (p3p.cpp, solvepnp.cpp).
 * samples: H x m int32 indices into the correspondence arrays (m = 7, 5, 4).
 * ---------------------------------------------------------------------------------- */

/* models H x 3 x 9 f64, nmodels H int32 (0..3). */
int vo_solve_7pt(const int32_t* samples, int H, const float* pts, int N,
                 double* models, int32_t* nmodels, void* stream);
/* models H x 10 x 9 f64, nmodels H int32 (0..10).  pts N x 4 f64 K-normalised. */
int vo_solve_5pt(const int32_t* samples, int H, const double* pts, int N,
                 double* models, int32_t* nmodels, void* stream);
/* models H x 12 f64 (R after the Rodrigues round trip OpenCV's model format implies, t),
 * rvecs H x 3 f64, nmodels H int32 (0 or 1). */
int vo_solve_p3p(const int32_t* samples, int H, const float* obj, const float* img,
                 const double* K4_host, int N, double* models, double* rvecs,
                 int32_t* nmodels, void* stream);

/* ------------------------------------------------------------------------------------
 * Best-model bookkeeping of RANSACPointSetRegistrator::run (ptsetreg.cpp): strict
 * "good > max(maxGood, modelPoints-1)" in iteration-then-model order, with the adaptive
 * iteration count of RANSACUpdateNumIters cutting the scan short.
 * counts: n_iters x max_models int32, nmodels: n_iters int32 (NULL = 1 each, or counts<0
 * marks a missing model).  result (4 x int32): best_iter (-1 none), best_model, max_good,
 * final niters.
 * ---------------------------------------------------------------------------------- */
int vo_ransac_replay(const int32_t* counts, const int32_t* nmodels, int n_iters, int max_models,
                     int n_points, int model_points, double confidence, int max_iters,
                     int32_t* result, void* stream);
/* Whole RANSAC runs for `batch` problems of one shape, as four launches in total (solve, score,
 * replay, mask) with no host round trip in between: the device-resident form of
 *   kind 0: cv2.findFundamentalMat(..., cv2.RANSAC, ...)    reference src/utils/geometry.py:39
 *   kind 1: cv2.findEssentialMat(..., cv2.RANSAC, ...)      reference src/utils/geometry.py:44
 *   kind 2: cv.solvePnPRansac(..., SOLVEPNP_P3P, ...)       reference src/pnp.py:16-26 (without the
 *           final EPnP refit, which is vo_host_epnp on the inliers)
 * samples  batch x H x m int32 (m = 7 / 5 / 4), the getSubset stream of each problem
 * pts      kind 0: batch x N x 4 f32; kind 1: batch x N x 4 f64 (K-normalised); kind 2: batch x N x 3 f32
 *          object points, with img batch x N x 2 f32 and K4_host (img, K4_host unused otherwise)
 * n_dev    optional batch x int32 on the device: correspondences actually present in each problem
 *          (<= N, the array stride); NULL = N everywhere
 * result   batch x 4 int32: best iteration (-1 none), best model, maxGood, final niters
 * best_model batch x 9 (kinds 0, 1) or batch x 12 ([R|t], kind 2) f64; best_rvec batch x 3 f64 (kind 2,
 *          may be NULL); mask batch x N uint8.  Scoring skips the empty model slots of samples with
 *          fewer solutions than the maximum.
 * first_block  0: all H iterations are solved and scored in one go.  > 0: iterations run in blocks
 *          ([0, first_block), then blocks four times as long as everything before); after each block
 *          the replay knows which problems OpenCV's adaptive loop would already have left (niters
 *          <= iterations done) and their remaining iterations are skipped -- same result, less work
 *          when inliers are plentiful, a few more launches when they are not. */
size_t vo_ransac_workspace_bytes(int kind, int H, int batch);
int vo_ransac_batch(int kind, const int32_t* samples, int H, const void* pts, const float* img,
                    const double* K4_host, int N, const int32_t* n_dev, int batch, float thr2,
                    double confidence, int max_iters, int first_block, int32_t* result,
                    double* best_model, double* best_rvec, uint8_t* mask, void* workspace,
                    size_t workspace_bytes, void* stream);

/* The small-input branches of cv2.findFundamentalMat(..., cv2.RANSAC, ...) (reference src/utils/geometry.py:39 with
 * few matches) for a whole batch of frame pairs, on the device: pairs with 8..14 correspondences get
 * OpenCV's LMedS registrator (getSubset with its default 1000 attempts, 7-point models ranked by the median of
 * their errors, the first strictly smallest wins, inliers within 2.5 * 1.4826 * (1 + 5/(n-7)) * sqrt(median), at
 * least 7 of them or no model), pairs with exactly 7 get every solution of the seven points as they stand
 * (models7 batch x 3 x 9, nmodels7 batch; F = the first solution, mask all ones -- cv2 returns the solutions
 * stacked as a 3k x 3 matrix).  Rows of result / F / mask of pairs with other counts are left untouched, so the
 * outputs of vo_ransac_batch(kind 0) can be passed in directly.  n_match: batch x int32 true correspondence counts;
 * pts4 batch x N x 4 f32.  The number of LMedS iterations follows from confidence and max_iters
 * (vo_fund_small_iterations); result rows hold it as their fourth entry. */
int vo_fund_small_iterations(double confidence, int max_iters);
size_t vo_fund_small_workspace_bytes(double confidence, int max_iters, int batch);
int vo_fund_small_batch(const float* pts4, int N, const int32_t* n_match, int batch, double confidence, int max_iters,
                        int32_t* result, double* F, uint8_t* mask, double* models7, int32_t* nmodels7,
                        void* workspace, size_t workspace_bytes, void* stream);

/* One vo_ransac_batch over several GPUs (the hypothesis range of every problem is split over
 * the ranks, correspondences are replicated).  No counterpart in the reference, which is single-process;
 * the three calls reproduce vo_ransac_batch(first_block = 0) bit for bit:
 *   1. vo_ransac_slice_counts  every rank: solve + score iterations [h_lo, h_hi) of all `batch` problems into
 *      its workspace (same size and layout as vo_ransac_batch's; vo_ransac_layout gives the byte offsets of
 *      the nmodels table, batch x H int32, and of the counts table, batch x H x width int32, and width);
 *   2. the caller exchanges the [h_lo, h_hi) slices of both tables between the ranks (ONE all_gather for
 *      the whole batch; visual_odometry_amd/parallel.py) and writes the others' slices into its own tables;
 *   3. vo_ransac_slice_finish  every rank: the replay of OpenCV's sequential bookkeeping over the complete
 *      tables (replicated, so every rank gets the same `result`), then best_model / best_rvec = the
 *      winner's model where this rank solved it (best iteration in [h_lo, h_hi)) and all-zero bytes
 *      elsewhere, so that an integer SUM all-reduce of the raw 8-byte words leaves the winner's bits on
 *      every rank;
 *   4. vo_ransac_mask_models   mask[b] = inliers of best_model[b] (batch x 9 | 12 f64) for problems with
 *      result[4b] >= 0, zeros otherwise.
 * offsets_host: 4 x size_t = {nmodels offset, counts offset, width, total bytes}. */
int vo_ransac_layout(int kind, int H, int batch, size_t* offsets_host);
int vo_ransac_slice_counts(int kind, const int32_t* samples, int H, int h_lo, int h_hi, const void* pts,
                           const float* img, const double* K4_host, int N, const int32_t* n_dev, int batch,
                           float thr2, void* workspace, size_t workspace_bytes, void* stream);
int vo_ransac_slice_finish(int kind, int H, int h_lo, int h_hi, int N, const int32_t* n_dev, int batch,
                           double confidence, int max_iters, void* workspace, size_t workspace_bytes,
                           int32_t* result, double* best_model, double* best_rvec, void* stream);
int vo_ransac_mask_models(int kind, const double* best_model, const int32_t* result, const void* pts,
                          const float* img, const double* K4_host, int N, const int32_t* n_dev, int batch,
                          float thr2, uint8_t* mask, void* stream);

/* vo_ransac_batch for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments,
 * results and kernels as vo_ransac_batch, minus the workspace: the scratch memory (model slots,
 * compacted model lists, count tables) is a fixed array of VO_RANSAC_STATIC_MAX_BYTES inside the
 * library's code object, one copy per GPU, held until the process ends.  The call itself performs no
 * hipMalloc / hipFree or any other allocating call; inputs and outputs are still the caller's.
 *   - problems whose vo_ransac_workspace_bytes(kind, H, batch) exceeds vo_ransac_static_max_bytes()
 *     are rejected with VO_ESIZE (what fits, as H x batch: kind 0 up to 262144, e.g. 4096 iterations x 64
 *     pairs; kind 1 up to 65536; kind 2 up to 262144 -- ask vo_ransac_workspace_bytes for other shapes);
 *   - all calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 *     threads) race; serialise them or use vo_ransac_batch with separate workspaces;
 *   - the array keeps the previous call's contents; the call clears what must start at zero. */
#define VO_RANSAC_STATIC_MAX_BYTES (64u << 20)
size_t vo_ransac_static_max_bytes(void);
int vo_ransac_batch_static(int kind, const int32_t* samples, int H, const void* pts, const float* img,
                           const double* K4_host, int N, const int32_t* n_dev, int batch, float thr2,
                           double confidence, int max_iters, int first_block, int32_t* result,
                           double* best_model, double* best_rvec, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------
 * Keeping a frame pair on the device between matching and RANSAC.
 * ---------------------------------------------------------------------------------- */

/* The gathers of reference src/features.py:110-122 / src/twoDtwoD.py:54-55 as an ordered stream
 * compaction, `batch` pairs per launch.  idx batch x nq x 2 (vo_knn2_l2_i8), keep batch x nq
 * (vo_ratio_test), kp_q batch x nq x 2 f32 and kp_t batch x nt x 2 f32 keypoint positions, xyz_q
 * batch x nq x 3 f32 landmarks of the query keypoints (NULL if none).  Outputs, all batch x nq
 * rows with the first n_match[b] used, ascending query index: match_q / match_t (row indices),
 * pts4 (x_q y_q x_t y_t: the F / E correspondences), obj + img (landmark of the query keypoint and
 * position of its match: the PnP correspondences of src/sift_test.py:104-119).  pts4, obj, img may
 * each be NULL. */
int vo_match_compact(const int32_t* idx, const uint8_t* keep, int nq, int nt, int batch,
                     const float* kp_q, const float* kp_t, const float* xyz_q,
                     int32_t* match_q, int32_t* match_t, int32_t* n_match,
                     float* pts4, float* obj, float* img, void* stream);

/* vo_host_gen_subsets on the device, for correspondence counts that only exist there: n_dev batch
 * x int32 (NULL = N everywhere), N the stride of pts4 (batch x N x 4 f32; NULL = no checkSubset,
 * i.e. the E / PnP estimators).  samples batch x H x model_points; iterations after getSubset gave
 * up are filled with -1 (the solvers treat them as "no sample"), n_generated (batch, may be NULL)
 * receives how many were produced.  model_points 4, 5 or 7. */
int vo_gen_subsets_device(const int32_t* n_dev, int N, int model_points, int H, const float* pts4,
                          int batch, int32_t* samples, int32_t* n_generated, void* stream);

/* ------------------------------------------------------------------------------------
 * The O(N) tail of the 2D-2D initialisation.
 * ---------------------------------------------------------------------------------- */

/* cv.triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2) -- reference
 * src/utils/disambiguate_relative_pose.py:42, src/twoDtwoD.py:110, src/tracking.py:153.
 * P1_host / P2_host: 3 x 4 row-major f64 on the host; x1, x2: N x 2 f64 pixel positions on the
 * device (one row per point); X4: N x 4 homogeneous points, un-normalised (scale and sign as the
 * SVD leaves them, like OpenCV). */
int vo_triangulate(const double* P1_host, const double* P2_host, const double* x1, const double* x2,
                   int N, double* X4, void* stream);

/* For `batch` frame pairs straight from vo_ransac_batch(kind 0): E = K^T F K (reference
 * src/utils/geometry.py:48-49), decomposeEssentialMatrix (src/utils/decompose_essential_matrix.py:4-37)
 * and disambiguateRelativePose (src/utils/disambiguate_relative_pose.py:5-72) over the inliers in
 * `mask`.  F batch x 9; result batch x 4 (pairs with result[4b] < 0 have no model and get the
 * identity; NULL = all have one); mask batch x N; pts4 batch x N x 4 f32 (x_i y_i x_j y_j); n_dev
 * per-pair counts or NULL; K_host 3 x 3 row-major (both cameras).  Outputs: cand batch x 4 x 12 the
 * four [R|t] candidates in the reference's order (R0,+t R0,-t R1,+t R1,-t); score batch x 4 their
 * cheirality scores; T batch x 16 = T_cami_camj of the first strict maximum; X batch x N x 4 the
 * dehomogenised points in camera-i coordinates (zero rows where mask is 0); keep batch x N =
 * mask & depth > min(0, T[2][2]) & |X| < max_depth.  The 3 x 3 SVD is a Jacobi SVD here and
 * LAPACK in the reference: same candidate set, possibly in another order (see vo_pose.hip). */
int vo_pose_tail_batch(const double* F, const int32_t* result, const uint8_t* mask, const float* pts4,
                       const int32_t* n_dev, const double* K_host, int N, int batch, double max_depth,
                       double* cand, int32_t* score, double* T, double* X, uint8_t* keep, void* stream);

/* The EPnP refit cv.solvePnPRansac runs on the inliers of the winning P3P model (reference
 * src/pnp.py:16-26: the pose `pnp` returns), for `batch` pairs straight from vo_ransac_batch(kind 2):
 * obj batch x N x 3 f32, img batch x N x 2 f32, mask batch x N (the PnP inlier masks), n_dev per-pair
 * counts or NULL, result batch x 4 or NULL (pairs with result[4b] < 0 have no model), K4_host =
 * fx fy cx cy.  Outputs: Rt batch x 12 f64 (R row-major, then t; world -> camera as OpenCV returns
 * it), ok batch (1 = pose, 0 = no model / fewer than 4 inliers / degenerate geometry, Rt zeroed).
 * Device counterpart of vo_host_epnp; agrees with it to rounding (block reductions instead of
 * sequential sums). */
int vo_epnp_batch(const float* obj, const float* img, const uint8_t* mask, const int32_t* n_dev,
                  const int32_t* result, const double* K4_host, int N, int batch, double* Rt, int32_t* ok,
                  void* stream);

/* The ranking step of OpenCV's LMedS registrator, which cv2.findFundamentalMat(..., cv2.RANSAC, ...)
 * silently uses for 8..14 points (reference src/utils/geometry.py:39 with few matches):
 * medians[h] = the f32 error of rank N/2 of model h over the N correspondences (errors ordered as
 * their int reinterpretation, as std::nth_element over errf.ptr<int>() does); +Inf where valid[h]
 * is 0.  models H x 9, pts N x 4 f32 (x1 y1 x2 y2), N <= 64. */
int vo_lmeds_median_fund(const double* models, const uint8_t* valid, int H, const float* pts, int N,
                         float* medians, void* stream);

/* result (2 x int32): largest count and the lowest index holding it (pure argmax mode). */
int vo_argmax_first(const int32_t* counts, int n, int32_t* result, void* stream);

/* ------------------------------------------------------------------------------------
 * Host-only helpers (no GPU): the sample stream of RANSACPointSetRegistrator::getSubset
 * driven by cv::RNG((uint64)-1), incl. FMEstimatorCallback::checkSubset when pts4_host
 * (N x 4 f32: x1 y1 x2 y2) is given.  Returns the number of iterations generated (the
 * stream stops at the first iteration whose 10000 attempts all fail), or a VO_E* code.
 * ---------------------------------------------------------------------------------- */
int vo_host_gen_subsets(int n_points, int model_points, int max_iters,
                        const float* pts4_host, int32_t* subsets_host);
/* The same with getSubset's maxAttempts spelled out: RANSACPointSetRegistrator::run passes 10000 (what
 * vo_host_gen_subsets uses), LMeDSPointSetRegistrator::run -- cv2.findFundamentalMat with 8..14 points -- leaves
 * the default of 1000. */
int vo_host_gen_subsets_attempts(int n_points, int model_points, int max_iters, int max_attempts,
                                 const float* pts4_host, int32_t* subsets_host);
int vo_host_update_num_iters(double p, double ep, int model_points, int max_iters);

/* Host-only: the EPnP refit cv.solvePnPRansac(flags=SOLVEPNP_P3P) applies to the inliers of the
 * winning model (reference src/pnp.py:16-26; OpenCV calib3d epnp.cpp).  obj_host n x 3 f64,
 * img_host n x 2 f64 pixels, K4_host = {fx, fy, cx, cy}; R9_host row-major, t3_host: world ->
 * camera.  Returns 1 on success, 0 when no finite pose comes out, or a VO_E* code. */
int vo_host_epnp(const double* obj_host, const double* img_host, int n, const double* K4_host,
                 double* R9_host, double* t3_host);

/* ------------------------------------------------------------------------------------
 * Synthetic inputs (no counterpart in the reference, which reads datasets and runs cv2.SIFT: neither exists
 * here).  Generators of the offline sweep (visual_odometry_amd/sweep.py), HBM-write bound.  Every value is a pure
 * function of (keys[b], stream_id, element index), so a pair's data does not depend on its batch.
 * vo_synth_sift_like: out (batch x n x 128 f32) = SIFT-like integer rows (Gamma(0.6)-like magnitudes, normalise,
 * clip 0.2, renormalise, x512, round, saturate).  vo_synth_noisy_copies: train[b, dst[b,m]] = saturate(round(
 * query[b, src[b,m]] + sigma N(0,1))) for m < nmatch; src / dst are batch x nmatch int64 row numbers (rows outside
 * their set are skipped).  keys: batch int64 on the device.
 * ---------------------------------------------------------------------------------- */
int vo_synth_sift_like(const int64_t* keys, int batch, int n, uint64_t stream_id, float* out, void* stream);
/* out (batch x n int64) = the raw hash words behind the draws: splitmix64's finaliser of
 * keys[b] ^ mix(i * 0x9E3779B97F4A7C15 + mix(stream_id)). */
int vo_synth_hash64(const int64_t* keys, int batch, long n, uint64_t stream_id, int64_t* out, void* stream);
int vo_synth_noisy_copies(const int64_t* keys, int batch, const float* query, int nq, const int64_t* src,
                          const int64_t* dst, int nmatch, uint64_t stream_id, float sigma, float* train, int nt,
                          void* stream);

/* ------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference): HIP-event timing of single launches inside
 * the entry points above, on the stream they run on.  stage_mask bits: 0 solve, 1 score, 2 replay,
 * 3 mask (inside vo_ransac_batch; tag = kind | iteration-block index << 8), 4 the kNN sweep kernel
 * (inside vo_knn2_l2_i8; tag 0x1000).
 * vo_profile_enable(0, 0) switches it off.  vo_profile_collect waits for the recorded events and
 * returns how many (tag, milliseconds) records it wrote; it resets the record list.
 * ---------------------------------------------------------------------------------- */
int vo_profile_enable(int stage_mask, int capacity);
int vo_profile_collect(int32_t* tags_host, float* ms_host, int max_records);

#ifdef __cplusplus
}
#endif
#endif
