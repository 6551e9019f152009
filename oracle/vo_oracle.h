/*
 * vo_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic that the reference (Heusini/visual_odometry)
 * delegates to OpenCV 4.6.0 (opencv-python==4.6.0.66, reference vision.yml:17) on its
 * hot path:
 *   - cv.BFMatcher().knnMatch(q, t, k=2)          reference src/features.py:100-101
 *   - cv2.findFundamentalMat(..., cv2.RANSAC,...)  reference src/utils/geometry.py:39
 *   - cv2.findEssentialMat(..., cv2.RANSAC, ...)   reference src/utils/geometry.py:44
 *   - cv.solvePnPRansac(..., SOLVEPNP_P3P, ...)    reference src/pnp.py:16-26
 *   - cv.Rodrigues                                 reference src/pnp.py:32
 *   - cv.triangulatePoints                         reference src/utils/disambiguate_relative_pose.py:42
 *
 * OpenCV itself is NOT present in this build environment (no cv2 module, no headers,
 * no source), and the reference ships no tests, golden vectors or seeds (SURVEY.md 4,
 * 8c).  Every routine here is therefore written from the published algorithms
 *  and **PARITY IS UNPINNED** against real OpenCV.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (visual_odometry_amd/) never links, imports or calls it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- global numeric mode -------------------------------------------------------- */
/* 0 (default): libm acos/cos/pow/hypot/sin exactly where OpenCV calls them.
 * 1: deterministic "det-math" replacements built from IEEE + - * / sqrt only, in the
 *    same operation order as the HIP device code, so oracle and GPU agree bit for bit. */
void vo_or_set_det_math(int on);
int  vo_or_get_det_math(void);

/* ---- A.1 brute-force 2-NN ------------------------------------------------------- */
/* q: nq x dim f32, t: nt x dim f32.  idx: nq x 2 (i32, -1 when missing), dist: nq x 2
 * f32 (FLT_MAX when missing).  f32 accumulation + sqrtf, strict '<' insertion. */
int vo_or_knn2(const float* q, int nq, const float* t, int nt, int dim,
               int32_t* idx, float* dist);

/* ---- A.2 cv::RNG / getSubset / RANSAC bookkeeping ------------------------------- */
void vo_or_rng_stream(uint64_t seed, int n, uint32_t* out);
/* check_mode 0: none; 1: fundamental-matrix collinearity test on m1,m2 (N x 2 f32).
 * Generates subsets for iterations 0..max_iters-1 from RNG((uint64)-1); stops at the
 * first iteration whose 10000 attempts all fail.  Returns #iterations generated. */
int vo_or_gen_subsets_attempts(int count, int model_points, int max_iters, int check_mode,
                               const float* m1, const float* m2, int32_t* subsets, int max_attempts);
int vo_or_gen_subsets(int count, int model_points, int max_iters, int check_mode,
                      const float* m1, const float* m2, int32_t* subsets);
int vo_or_update_num_iters(double p, double ep, int model_points, int max_iters);
/* counts: n_gen x max_models, nmodels: n_gen.  Replays RANSACPointSetRegistrator::run
 * over pre-computed inlier counts.  out[0]=best iter (-1 none) out[1]=best model k
 * out[2]=maxGood out[3]=final niters. */
void vo_or_ransac_replay(const int32_t* counts, const int32_t* nmodels, int n_gen,
                         int max_models, int count, int model_points,
                         double confidence, int max_iters, int32_t* out);
int vo_or_find_inliers(const float* err, int n, double thresh, uint8_t* mask);

/* ---- A.3 fundamental matrix ----------------------------------------------------- */
void vo_or_fund_error(const double* F, const float* m1, const float* m2, int n, float* err);
int  vo_or_fund_check_subset(const float* ms1, const float* ms2, int count);
/* 7 correspondences (f32 x,y) -> up to 3 F (row-major 9 doubles each).  returns n. */
int  vo_or_run7point(const float* ms1, const float* ms2, double* F27, int hartley);
/* whole findFundamentalMat(...,FM_RANSAC,...).  returns 1 ok / 0 no model.
 * F9 out, mask n out, info[0]=best iter info[1]=k info[2]=maxGood info[3]=niters */
int  vo_or_find_fundamental_ransac(const float* m1, const float* m2, int n, double thr,
                                   double conf, int max_iters, int hartley,
                                   double* F9, uint8_t* mask, int32_t* info);
/* the LMedS registrator OpenCV 4.6.0 uses for 8..14 points (A.3) */
int  vo_or_find_fundamental_lmeds(const float* m1, const float* m2, int n, double conf, int max_iters,
                                  int hartley, double* F9, uint8_t* mask, int32_t* info);

/* ---- A.4 essential matrix ------------------------------------------------------- */
void vo_or_sampson_error(const double* E, const double* x1, const double* x2, int n, float* err);
int  vo_or_run5point(const double* q1, const double* q2, double* E90);
int  vo_or_find_essential_ransac(const double* p1, const double* p2, int n, const double* K9,
                                 double prob, double thr, int max_iters,
                                 double* E9, uint8_t* mask, int32_t* info);

/* ---- A.5/A.6 PnP ---------------------------------------------------------------- */
void vo_or_rodrigues_v2m(const double* rvec, double* R9);
void vo_or_rodrigues_m2v(const double* R9, double* rvec);
/* R row-major, zero distortion.  out: n x 2 doubles (pixels before f32 rounding) */
void vo_or_project_points(const double* R9, const double* t3, const double* K4,
                          const float* obj, int n, double* out);
void vo_or_pnp_error(const double* R9, const double* t3, const double* K4,
                     const float* obj, const float* img, int n, float* err);
/* 4 correspondences -> best (rvec,tvec,R).  returns #solutions (0 => none). */
int  vo_or_p3p_kernel(const float* obj4, const float* img4, const double* K4,
                      double* rvec3, double* tvec3, double* R9);
int  vo_or_epnp(const double* obj, const double* img, int n, const double* K4,
                double* R9, double* t3);
int  vo_or_solve_pnp_ransac(const float* obj, const float* img, int n, const double* K4,
                            int max_iters, float reproj_err, double conf,
                            double* rvec3, double* tvec3, uint8_t* mask, int32_t* info);

/* ---- A.7 triangulation ---------------------------------------------------------- */
void vo_or_triangulate(const double* P1, const double* P2, const double* x1,
                       const double* x2, int n, double* X4);

/* ---- helpers exposed for tests -------------------------------------------------- */
void vo_or_jacobi_svd(double* At, int m, int n, int n1, double* W, double* Vt);
int  vo_or_solve_cubic(const double* c4, double* roots3);
int  vo_or_solve_deg4(double a, double b, double c, double d, double e, double* roots4);
int  vo_or_solve_poly(const double* coeffs, int deg, double* re, double* im);

#ifdef __cplusplus
}
#endif
#endif
