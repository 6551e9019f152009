// vo_score.hip -- RANSAC hypothesis scoring on gfx950: for H models x N correspondences,
// the error OpenCV's computeError callbacks define and the inlier count findInliers derives
// from it.  Stands in for the inner loop of cv2.findFundamentalMat / cv2.findEssentialMat /
// cv.solvePnPRansac (reference src/utils/geometry.py:39,44 and src/pnp.py:16-26; semantics:
// SURVEY.md Appendix A.2-A.6, Appendix B for the exactness rules).
//
// Exactness: counts and masks are bit for bit those of the scalar definitions below (fund_err, sampson_err,
// pnp_err: the same IEEE f64/f32 operations in the same order as a CPU evaluation; this file is compiled with
// -ffp-contract=off; f64 divide and f32<->f64 conversions are correctly rounded on gfx950).  The COUNT kernels
// for F and PnP do not evaluate those definitions for every correspondence: a division-free f32 filter with
// rigorous error bounds settles every verdict that is not within rounding distance of the threshold ("sure
// inlier" / "sure outlier", left in sign bits), and only the rest takes the f64 filter (F) and the exact
// evaluation.  The mask kernels always run the exact definitions.
//
// Mapping: one thread owns one model (registers); a workgroup of 256 models walks a chunk of <= 512
// correspondences staged in LDS (f32 for F and PnP, f64 for E), so the inner loop reads every operand with
// wave-uniform (broadcast) ds_read and is pure VALU.  The grid is (model blocks) x (point chunks) x (problems);
// chunk sums meet in counts[] through integer atomics, which are order-independent.
#include "vo_common.h"

#ifndef VO_SCORE_G
#define VO_SCORE_G 8                  // unroll factor of the tier-1 loops of the F count kernel (2: 5.25 ms, 4: 4.96, 8: 4.86, 16: 5.30)
#endif
#ifndef VO_SCORE_GP
#define VO_SCORE_GP 4                 // unroll factor of the PnP / E filter loops
#endif
#ifndef VO_SCORE_MINW
#define VO_SCORE_MINW 1               // launch-bounds hint: waves per SIMD the register budget must allow
#endif
#ifndef VO_SCORE_MINW_F
#define VO_SCORE_MINW_F 5             // the F count kernel: 5 waves per SIMD (96 VGPRs; the f64 tiers are a call and spill around it)
#endif
#ifndef VO_SCORE_NE
#define VO_SCORE_NE 2                 // tier 1b: queued pairs each lane works on side by side
#endif
#ifndef VO_SCORE_CONST_LDS
#define VO_SCORE_CONST_LDS 1          // 1: tier 1b reads the models' f32 filter constants from LDS (12 KB) instead of lane shuffles
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunk = 512;          // points per workgroup pass (16 KB of LDS at 32 B each)

// ---- error definitions (shared by the count and the mask kernels) ----------------------

// FMEstimatorCallback::computeError: max of the two squared point-to-epipolar-line distances
__device__ __forceinline__ float fund_err(const double* F, double x1, double y1, double x2, double y2)
{
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * x1 + F[1] * y1 + F[2];
    b = F[3] * x1 + F[4] * y1 + F[5];
    c = F[6] * x1 + F[7] * y1 + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 < e2 ? e2 : e1);
}

// Sure-inlier / sure-outlier filter for fund_err, used by the count kernel only.  The verdict
// "(float)max(e1,e2) <= thr2" needs two correctly rounded divides per correspondence; they decide
// nothing unless d^2 and thr2*den are within rounding distance of each other.  The filter evaluates
// the two residuals with fused multiply-adds (any evaluation order is fine here) and returns
//   +1  both residuals are below the threshold by more than every rounding error that can separate
//       this evaluation from fund_err's,
//   -1  one residual is above it by more than that,
//    0  cannot tell (includes every NaN/Inf/zero-denominator case): the caller runs fund_err.
// Error budget (eps = 2^-53; Q1 >= max(|x1|,|y1|,1) and Q2 likewise over the staged chunk; q = (Q,Q,1)):
//   S  = sum_ij |F_ij| q2_i q1_j bounds every partial sum of x2^T F x1, so the two evaluations of d
//        differ by <= 12 eps S                                   (eta carries 2^-44 S);
//   al = max row (column) sum of |F| weighted by q bounds the same for a and b: <= 6 eps al apart
//        (alpha carries 2^-46 al), hence den by <= 2(|a|+|b|) alpha + 2 alpha^2 + 4 eps den
//        <= 2^-21 den + (2^22+2) alpha^2   (AM-GM with (|a|+|b|)^2 <= 2 den);
//   the divide, the product and the f32 cast of the exact path move e by < 2^-22 relative.
// With z = d^2 - thr2 den and
//   m = 2 eta |d| + 2^-19 thr2 den + 2 (eta^2 + 2^24 thr2 alpha^2) + 2^-600:
//   z < -m  =>  (|d|+eta)^2 <= thr2 (1-2^-20)(den-dlt)   and   z > m  =>  (|d|-eta)^2 >= thr2 (1+2^-20)(den+dlt).
// A NaN anywhere makes both compares false; an Inf makes m Inf or NaN: both end in "cannot tell".
struct FundFilter {
    double t, rel, eta2, c1, c2;       // thr2, 2^-19 thr2, 2 eta, constant terms of m for den1 / den2
};

__device__ __forceinline__ FundFilter fund_filter_setup(const double* F, double thr2, double Q1, double Q2)
{
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = fabs(F[k]);
    const double S = (A[0] + A[1] + A[3] + A[4]) * Q2 * Q1 + (A[2] + A[5]) * Q2 + (A[6] + A[7]) * Q1 + A[8];
    const double eta = 0x1p-44 * S;
    // den2 = a^2+b^2 of F x1 (rows 0,1 of F); den1 of F^T x2 (columns 0,1 of F)
    const double al2 = 0x1p-46 * fmax((A[0] + A[1]) * Q1 + A[2], (A[3] + A[4]) * Q1 + A[5]);
    const double al1 = 0x1p-46 * fmax((A[0] + A[3]) * Q2 + A[6], (A[1] + A[4]) * Q2 + A[7]);
    FundFilter f;
    f.t = thr2;
    f.rel = 0x1p-19 * thr2;
    f.eta2 = 2. * eta;
    f.c1 = 2. * (eta * eta + 0x1p24 * thr2 * al1 * al1) + 0x1p-600;
    f.c2 = 2. * (eta * eta + 0x1p24 * thr2 * al2 * al2) + 0x1p-600;
    return f;
}

__device__ __forceinline__ int fund_filter(const double* F, const FundFilter& f, double x1, double y1, double x2, double y2)
{
    const double a1 = __builtin_fma(F[0], x1, __builtin_fma(F[1], y1, F[2]));
    const double b1 = __builtin_fma(F[3], x1, __builtin_fma(F[4], y1, F[5]));
    const double c1 = __builtin_fma(F[6], x1, __builtin_fma(F[7], y1, F[8]));
    const double den2 = __builtin_fma(a1, a1, b1 * b1);
    const double d2 = __builtin_fma(x2, a1, __builtin_fma(y2, b1, c1));
    const double a2 = __builtin_fma(F[0], x2, __builtin_fma(F[3], y2, F[6]));
    const double b2 = __builtin_fma(F[1], x2, __builtin_fma(F[4], y2, F[7]));
    const double c2 = __builtin_fma(F[2], x2, __builtin_fma(F[5], y2, F[8]));
    const double den1 = __builtin_fma(a2, a2, b2 * b2);
    const double d1 = __builtin_fma(x1, a2, __builtin_fma(y1, b2, c2));
    const double z1 = __builtin_fma(-f.t, den1, d1 * d1), z2 = __builtin_fma(-f.t, den2, d2 * d2);
    const double m1 = __builtin_fma(f.eta2, fabs(d1), __builtin_fma(f.rel, den1, f.c1));
    const double m2 = __builtin_fma(f.eta2, fabs(d2), __builtin_fma(f.rel, den2, f.c2));
    if ((z1 < -m1) & (z2 < -m2)) return 1;
    if ((z1 > m1) | (z2 > m2)) return -1;
    return 0;
}

// First stage of the same filter in f32 (wave64 f32 FMAs issue ~1.7x faster than f64 ones on
// gfx950 and the verdict of most correspondences is nowhere near the threshold).  The model is
// scaled by a power of two so that max|F_k| lies in [0.5,1) (the verdict is scale invariant and the
// scaling is exact), which keeps every quantity that matters far from f32 underflow, then rounded
// to f32.  Budget with u = 2^-24: model rounding + two FMAs put a,b,c within 3u of their partial-sum
// bound, d within 5u S (eta carries 2^-20 S), den within 2(|a|+|b|) alpha + 2 alpha^2 + 2u den with
// alpha = 2^-21 al  =>  <= 2^-19 den + 2^21 alpha^2 by AM-GM; z and m themselves are rounded to f32
// (<= 2^-22 relative of d^2 + thr2 den); the exact path's own roundings are < 2^-22.  All of it is
// covered, with at least 2x to spare per term, by
//   m = 2 eta |d| + 2^-17 thr2 den + 2 (eta^2 + 2^22 thr2 alpha^2).
// A model that is zero, NaN, Inf or scaled beyond 2^+-300 gets eta = Inf: every verdict is "cannot tell".
struct FundFilter32 {
    float F[9];
    float tlo, thi, eta2, c;           // thr2 -+ 2^-17 thr2 (rounded outwards), 2 eta, constant term of m
};

template <bool SAMPSON>
__device__ __forceinline__ FundFilter32 filter32_setup(const double* F, float thr2, double Q1, double Q2)
{
    FundFilter32 f;
    double fm = 0.;
#pragma unroll
    for (int k = 0; k < 9; ++k) fm = fmax(fm, fabs(F[k]));
    double nanprobe = 0.;
#pragma unroll
    for (int k = 0; k < 9; ++k) nanprobe += F[k] * 0.;           // NaN if any entry is NaN or Inf
    // The rescaled f32 evaluation cannot see the underflow/overflow the exact f64 evaluation runs into
    // for absurdly scaled models or thresholds; inside this window a sure verdict here (which needs
    // max(d^2, thr2 den) > 2^-41 in scaled units) keeps every exact-path quantity that decides in
    // the normal range.
    const bool usable = fm > 0x1p-300 && fm < 0x1p300 && thr2 >= 0x1p-100f && thr2 <= 0x1p100f && nanprobe == 0.;
    int e = 0;
    (void)frexp(usable ? fm : 1., &e);
    const double sigma = ldexp(1., -e);                          // fm * sigma in [0.5, 1)
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double g = F[k] * sigma;
        f.F[k] = (float)g;
        A[k] = fabs(g);
    }
    const double S = (A[0] + A[1] + A[3] + A[4]) * Q2 * Q1 + (A[2] + A[5]) * Q2 + (A[6] + A[7]) * Q1 + A[8];
    // SAMPSON (essential matrix on K-normalised f64 coordinates): the coordinates are rounded to f32 as well, one more
    // 2^-24 per factor of every term, and the denominator is the SUM of the four squares -- eta and alpha doubled
    const double eta = (SAMPSON ? 0x1p-19 : 0x1p-20) * S;
    const double al2 = (SAMPSON ? 0x1p-20 : 0x1p-21) * fmax((A[0] + A[1]) * Q1 + A[2], (A[3] + A[4]) * Q1 + A[5]);
    const double al1 = (SAMPSON ? 0x1p-20 : 0x1p-21) * fmax((A[0] + A[3]) * Q2 + A[6], (A[1] + A[4]) * Q2 + A[7]);
    const double up = 1. + 0x1p-20;                              // constants are rounded to f32: round up
    const float inf = __builtin_huge_valf();
    const double al = fmax(al1, al2);
    f.tlo = (float)((double)thr2 * (1. - 0x1p-17) * (1. - 0x1p-20));
    f.thi = (float)((double)thr2 * (1. + 0x1p-17) * up);
    f.eta2 = usable ? (float)(2. * eta * up) : inf;
    f.c = usable ? (float)(2. * (eta * eta + (SAMPSON ? 0x1p23 : 0x1p22) * (double)thr2 * al * al) * up) : inf;
    return f;
}

__device__ __forceinline__ FundFilter32 fund_filter32_setup(const double* F, float thr2, double Q1, double Q2)
{
    return filter32_setup<false>(F, thr2, Q1, Q2);
}

// d1 and d2 of fund_err are two roundings of the same number x2^T F x1, and eta bounds the distance
// of ANY evaluation of it from either, so one d serves both residuals; with c = max(c1, c2) the two
// tests collapse onto the smaller denominator:
//   sure inlier   <=>  d^2 + (eta2 |d| + c) < (thr2 - rel) min(den1, den2)
//   sure outlier  <=>  d^2 - (eta2 |d| + c) > (thr2 + rel) min(den1, den2)
__device__ __forceinline__ int fund_filter32(const FundFilter32& f, float x1, float y1, float x2, float y2)
{
    const float* F = f.F;
    const float a1 = __builtin_fmaf(F[0], x1, __builtin_fmaf(F[1], y1, F[2]));
    const float b1 = __builtin_fmaf(F[3], x1, __builtin_fmaf(F[4], y1, F[5]));
    const float c1 = __builtin_fmaf(F[6], x1, __builtin_fmaf(F[7], y1, F[8]));
    const float den2 = __builtin_fmaf(a1, a1, b1 * b1);
    const float d = __builtin_fmaf(x2, a1, __builtin_fmaf(y2, b1, c1));
    const float a2 = __builtin_fmaf(F[0], x2, __builtin_fmaf(F[3], y2, F[6]));
    const float b2 = __builtin_fmaf(F[1], x2, __builtin_fmaf(F[4], y2, F[7]));
    const float den1 = __builtin_fmaf(a2, a2, b2 * b2);
    const float dmin = fminf(den1, den2);
    const float mm = __builtin_fmaf(f.eta2, fabsf(d), f.c);
    const float lo = __builtin_fmaf(d, d, mm), hi = __builtin_fmaf(d, d, -mm);
    if (lo < f.tlo * dmin) return 1;
    if (hi > f.thi * dmin) return -1;
    return 0;
}

// The same two tests with the verdicts left in SIGN BITS instead of compare results: on gfx950 a v_cmp_f32
// costs 4.8 issue cycles and a v_cndmask 12 against 1.95 for a v_fma_f32 (scripts/ubench/clock.hip), so the
// compare-and-select bookkeeping of fund_filter32 costs as much as 18 of its FMAs.  Here
//   r_in  = lo - tlo dmin   (one FMA: the product is exact inside it)   sign bit set  <=>  sure inlier
//   r_out = thi dmin - hi                                               sign bit set  <=>  sure outlier
// and the caller shifts the two sign bits into per-lane bit strings (v_alignbit_b32).  An exact zero gives +0
// (sign clear: "cannot tell"), a negative value that underflows gives -0 (sign set, correctly).  The sign of a
// NaN is NOT reliable, so this form is only used where no NaN or Inf can arise: every coordinate of the
// staged chunk finite and <= 2^20 in magnitude, the lane's model usable (fund_filter32_setup) and thr2 within
// [2^-60, 2^40] -- then every intermediate is finite and far from f32 overflow (|a|,|b| <= 2^22, den <= 2^45,
// |d| <= 2^43, d^2 <= 2^86, thi dmin <= 2^86).  Everything else goes through fund_filter32.
__device__ __forceinline__ void fund_filter32_signs(const FundFilter32& f, float x1, float y1, float x2, float y2,
                                                    uint32_t& in_bits, uint32_t& out_bits)
{
    const float* F = f.F;
    const float a1 = __builtin_fmaf(F[0], x1, __builtin_fmaf(F[1], y1, F[2]));
    const float b1 = __builtin_fmaf(F[3], x1, __builtin_fmaf(F[4], y1, F[5]));
    const float c1 = __builtin_fmaf(F[6], x1, __builtin_fmaf(F[7], y1, F[8]));
    const float den2 = __builtin_fmaf(a1, a1, b1 * b1);
    const float d = __builtin_fmaf(x2, a1, __builtin_fmaf(y2, b1, c1));
    const float a2 = __builtin_fmaf(F[0], x2, __builtin_fmaf(F[3], y2, F[6]));
    const float b2 = __builtin_fmaf(F[1], x2, __builtin_fmaf(F[4], y2, F[7]));
    const float den1 = __builtin_fmaf(a2, a2, b2 * b2);
    const float dmin = fminf(den1, den2);
    const float mm = __builtin_fmaf(f.eta2, fabsf(d), f.c);
    const float lo = __builtin_fmaf(d, d, mm), nhi = __builtin_fmaf(-d, d, mm);   // nhi = -(d^2 - mm), same rounding
    const float r_in = __builtin_fmaf(-f.tlo, dmin, lo);
    const float r_out = __builtin_fmaf(f.thi, dmin, nhi);
    in_bits = __builtin_amdgcn_alignbit(in_bits, __float_as_uint(r_in), 31);       // (in_bits << 1) | sign(r_in)
    out_bits = __builtin_amdgcn_alignbit(out_bits, __float_as_uint(r_out), 31);
}

#if defined(VO_SCORE_STATS) || defined(VO_SCORE_MF_PROF) || defined(VO_SCORE_MF_CHECK)
__device__ unsigned long long g_stats[8];
extern "C" int vo_score_stats(unsigned long long* out8, int reset)
{
    if (reset) { unsigned long long z[8] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stats), 8 * sizeof(unsigned long long));
}
#endif
// First half of that test alone: min(den1, den2) <= den2, so "d^2 - mm > thi den2" already proves "sure outlier"
// and needs neither F^T x2 nor den1 -- 14 VALU instructions instead of 24.  It settles ~97 % of the evaluations
// of a RANSAC batch (hypotheses are mostly wrong, thresholds tight); the rest is taken through fund_filter32
// afterwards, per lane, from a bit string (score_kernel).  Same domain rules as fund_filter32_signs.
__device__ __forceinline__ void fund_filter32_outlier_sign(const FundFilter32& f, float x1, float y1, float x2, float y2,
                                                           uint32_t& out_bits)
{
    const float* F = f.F;
    const float a1 = __builtin_fmaf(F[0], x1, __builtin_fmaf(F[1], y1, F[2]));
    const float b1 = __builtin_fmaf(F[3], x1, __builtin_fmaf(F[4], y1, F[5]));
    const float c1 = __builtin_fmaf(F[6], x1, __builtin_fmaf(F[7], y1, F[8]));
    const float den2 = __builtin_fmaf(a1, a1, b1 * b1);
    const float d = __builtin_fmaf(x2, a1, __builtin_fmaf(y2, b1, c1));
    const float mm = __builtin_fmaf(f.eta2, fabsf(d), f.c);
    const float nhi = __builtin_fmaf(-d, d, mm);
    const float r_out = __builtin_fmaf(f.thi, den2, nhi);          // < 0: d^2 - mm > thi den2 >= thi min(den1, den2)
    out_bits = __builtin_amdgcn_alignbit(out_bits, __float_as_uint(r_out), 31);
}

// The Sampson distance of EMEstimatorCallback::computeError through the same kind of filter:
//   err = d^2 / (a1^2 + b1^2 + a2^2 + b2^2),  (a1 b1 c1) = E x1,  (a2 b2 .) = E^T x2,  d = x2 . (a1 b1 c1)
// evaluated in f32 on the f32-rounded coordinates.  With |d - d*| <= eta and every a, b within alpha of its exact
// value, the denominator is off by at most 4 alpha sqrt(den) + 4 alpha^2 + 2^-22 den
// <= (2^-19 + 2^-22) den + (2^21 + 4) alpha^2 (AM-GM), so
//   sure inlier   <=  d^2 + (2 eta |d| + c) < tlo den      sure outlier  <=  d^2 - (2 eta |d| + c) > thi den
// with c >= eta^2 + 2^22 thr2 alpha^2 (set up with 2x that) and tlo, thi = thr2 (1 -+ 2^-17) rounded outwards, which
// also covers the f32 rounding of the exact path's result and of this test.  Same domain rules as fund_filter32_signs.
__device__ __forceinline__ void ess_filter32_signs(const FundFilter32& f, float x1, float y1, float x2, float y2,
                                                   uint32_t& in_bits, uint32_t& out_bits)
{
    const float* F = f.F;
    const float a1 = __builtin_fmaf(F[0], x1, __builtin_fmaf(F[1], y1, F[2]));
    const float b1 = __builtin_fmaf(F[3], x1, __builtin_fmaf(F[4], y1, F[5]));
    const float c1 = __builtin_fmaf(F[6], x1, __builtin_fmaf(F[7], y1, F[8]));
    const float d = __builtin_fmaf(x2, a1, __builtin_fmaf(y2, b1, c1));
    const float a2 = __builtin_fmaf(F[0], x2, __builtin_fmaf(F[3], y2, F[6]));
    const float b2 = __builtin_fmaf(F[1], x2, __builtin_fmaf(F[4], y2, F[7]));
    const float den = __builtin_fmaf(a1, a1, __builtin_fmaf(b1, b1, __builtin_fmaf(a2, a2, b2 * b2)));
    const float mm = __builtin_fmaf(f.eta2, fabsf(d), f.c);
    const float lo = __builtin_fmaf(d, d, mm), nhi = __builtin_fmaf(-d, d, mm);
    const float r_in = __builtin_fmaf(-f.tlo, den, lo);
    const float r_out = __builtin_fmaf(f.thi, den, nhi);
    in_bits = __builtin_amdgcn_alignbit(in_bits, __float_as_uint(r_in), 31);
    out_bits = __builtin_amdgcn_alignbit(out_bits, __float_as_uint(r_out), 31);
}

// EMEstimatorCallback::computeError: Sampson distance on normalised coordinates
__device__ __forceinline__ float sampson_err(const double* E, double ax, double ay, double bx, double by)
{
    const double Ex1_0 = E[0] * ax + E[1] * ay + E[2] * 1.;
    const double Ex1_1 = E[3] * ax + E[4] * ay + E[5] * 1.;
    const double Ex1_2 = E[6] * ax + E[7] * ay + E[8] * 1.;
    const double Etx2_0 = E[0] * bx + E[3] * by + E[6] * 1.;
    const double Etx2_1 = E[1] * bx + E[4] * by + E[7] * 1.;
    const double x2tEx1 = bx * Ex1_0 + by * Ex1_1 + 1. * Ex1_2;
    const double a = Ex1_0 * Ex1_0, b = Ex1_1 * Ex1_1, c = Etx2_0 * Etx2_0, d = Etx2_1 * Etx2_1;
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

// PnPRansacCallback::computeError: projectPoints (f64, zero distortion) -> f32 pixels ->
// f32 squared distance
__device__ __forceinline__ float pnp_err(const double* M, double fx, double fy, double cx, double cy,
                                         double X, double Y, double Z, float iu, float iv)
{
    double x = M[0] * X + M[1] * Y + M[2] * Z + M[9];
    double y = M[3] * X + M[4] * Y + M[5] * Z + M[10];
    double z = M[6] * X + M[7] * Y + M[8] * Z + M[11];
    z = z != 0. ? 1. / z : 1.;
    x *= z;
    y *= z;
    const float pu = (float)(x * fx + cx), pv = (float)(y * fy + cy);
    const float dx = iu - pu, dy = iv - pv;
    float s = 0.f;
    s += dx * dx;
    s += dy * dy;
    return s;
}

// Sure-inlier / sure-outlier filter for pnp_err in f32, sign-bit form like fund_filter32_signs (the exact path
// costs ~180 issue cycles per correspondence on gfx950: 20 f64 operations, a correctly rounded f64 divide and two
// v_cvt_f32_f64 at 20 cycles each; this costs ~55).  With the rows of [R|t] scaled by fx, fy, 1 and by a common
// power of two (max entry in [0.5,1)), u = (iu - cx, iv - cy) and
//   e = z u - (x'', y'')     (x'' = fx x, y'' = fy y, z the three row products)
// the reprojection error is |e| / |z| in pixels, so "error <= T" is "|e|^2 <= T^2 z^2" -- no divide.  Budget
// (u = 2^-24; Sxy, Sz bound every partial sum of the pixel rows' and of the depth row's products over the staged
// chunk; Um, Im bound |u| and |(iu,iv)| over the chunk, all >= 1):
//   * f32 evaluation: x'', y'' within 2^-21 Sxy and z within zeta = 2^-21 Sz of their real values (entry rounding
//     + 3 FMAs: 2^-22 S, 2x slack), each component of e within a = 2^-20 (Sz Um + Sxy);
//   * the exact path against real arithmetic: as long as |z| >= 2^-10 Sz (no cancellation beyond 10 bits) its f64
//     noise is <= 2^-40 (Um + Im + 1 + err) + 2^-41 Sxy/Sz px, and the f32 rounding of the projected pixel and of
//     the f32 difference, squares and sum move sqrt(s) by <= 2^-20 err + 2 g0 with
//     g0 = 2^-23 (Im + Um + 2) + 2^-38 Sxy/Sz px;
//   * (|e| + sqrt2 a)^2 <= (1+rho) e^2 + 2 a^2 (1+1/rho) and (|z| - zeta)^2 >= (1-rho) z^2 - zeta^2 / rho
//     (AM-GM, rho = 2^-8), and the mirror images for the outlier side.
// Hence, with T = sqrt(thr2), T_lo = (T - 2 g0)(1 - 2^-20), T_hi = (T + 2 g0)/(1 - 2^-20), kappa = 2^-19 Sz^2:
//   sure inlier   <=  e2 + C < tlo (z2 - kappa)      tlo = T_lo^2 (1-rho)/(1+rho) (1 - 2^-20)
//   sure outlier  <=  e2 - C > thi z2  and  z2 > kappa   thi = T_hi^2 (1+rho)/(1-rho) (1 + 2^-20)
// (C collects the constant terms of both sides; z2 > kappa keeps the exact path's 1/z away from the z == 0
// special case and from cancellation).  Same domain rules as fund_filter32_signs: tame chunk, usable model,
// thr2 in [2^-40, 2^40]; everything else, and every "cannot tell", takes pnp_err.
struct PnpFilter32 {
    float R[12];                 // rows (fx r0 | fx t0), (fy r1 | fy t1), (r2 | t2), scaled, as R[0..2] t R[3] ...
    float tlo, thi, c, kappa;    // c already holds tlo * kappa for the inlier side
    bool usable;
};

__device__ __forceinline__ PnpFilter32 pnp_filter32_setup(const double* M, double fx, double fy, float thr2, double Pm,
                                                          double Um, double Im)
{
    PnpFilter32 f;
    double A[12];
    A[0] = fx * M[0]; A[1] = fx * M[1]; A[2] = fx * M[2]; A[3] = fx * M[9];
    A[4] = fy * M[3]; A[5] = fy * M[4]; A[6] = fy * M[5]; A[7] = fy * M[10];
    A[8] = M[6]; A[9] = M[7]; A[10] = M[8]; A[11] = M[11];
    double fm = 0., nanprobe = 0.;
#pragma unroll
    for (int k = 0; k < 12; ++k) { fm = fmax(fm, fabs(A[k])); nanprobe += A[k] * 0.; }
    f.usable = fm > 0x1p-300 && fm < 0x1p300 && nanprobe == 0. && thr2 >= 0x1p-40f && thr2 <= 0x1p40f;
    int e = 0;
    (void)frexp(f.usable ? fm : 1., &e);
    const double sigma = ldexp(1., -e);
    double Srow[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double row = 0.;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double g = A[4 * r + k] * sigma;
            f.R[4 * r + k] = (float)g;
            row += fabs(g) * (k < 3 ? Pm : 1.);
        }
        Srow[r] = row;
    }
    const double Sxy = fmax(Srow[0], Srow[1]), Sz = Srow[2];
    // a depth row that is (nearly) all zero against the pixel rows: 1/z amplifies the exact path's f64 noise beyond
    // what g0 budgets -- leave such poses to the exact path altogether
    f.usable = f.usable && Sz > 0. && Sxy <= 0x1p30 * Sz;
    const double rho = 0x1p-8, T = sqrt((double)thr2);
    const double g0 = 0x1p-23 * (Im + Um + 2.) + (f.usable ? 0x1p-38 * (Sxy / Sz) : 0.);
    const double Tlo = (T - 2. * g0) * (1. - 0x1p-20), Thi = (T + 2. * g0) / (1. - 0x1p-20);
    const double zeta = 0x1p-21 * Sz, a = 0x1p-20 * (Sz * Um + Sxy), kappa = 0x1p-19 * Sz * Sz;
    const double tlo = Tlo > 0. ? Tlo * Tlo * (1. - rho) / (1. + rho) * (1. - 0x1p-20) : -1.;     // -1: never sure inlier
    const double thi = Thi * Thi * (1. + rho) / (1. - rho) * (1. + 0x1p-20);
    const double c_in = (2. * a * a * (1. + 1. / rho) + Tlo * Tlo * zeta * zeta / rho) / (1. + rho) + fmax(tlo, 0.) * kappa;
    const double c_out = (2. * a * a / rho + Thi * Thi * zeta * zeta * (1. + 1. / rho)) / (1. - rho);
    f.tlo = (float)tlo;
    f.thi = (float)(thi * (1. + 0x1p-22));
    f.c = (float)(fmax(c_in, c_out) * (1. + 0x1p-20));
    f.kappa = (float)(kappa * (1. + 0x1p-22));
    return f;
}

__device__ __forceinline__ void pnp_filter32_signs(const PnpFilter32& f, float X, float Y, float Z, float ux, float uy,
                                                   uint32_t& in_bits, uint32_t& out_bits)
{
    const float* R = f.R;
    const float x = __builtin_fmaf(R[0], X, __builtin_fmaf(R[1], Y, __builtin_fmaf(R[2], Z, R[3])));
    const float y = __builtin_fmaf(R[4], X, __builtin_fmaf(R[5], Y, __builtin_fmaf(R[6], Z, R[7])));
    const float z = __builtin_fmaf(R[8], X, __builtin_fmaf(R[9], Y, __builtin_fmaf(R[10], Z, R[11])));
    const float ex = __builtin_fmaf(z, ux, -x), ey = __builtin_fmaf(z, uy, -y);
    const float e2 = __builtin_fmaf(ex, ex, ey * ey);
    const float z2 = z * z;
    const float lo = e2 + f.c, nhi = f.c - e2;
    const float r_in = __builtin_fmaf(-f.tlo, z2, lo);             // < 0: sure inlier (implies z2 > kappa)
    const float r_out = __builtin_fmaf(f.thi, z2, nhi);            // < 0: outlier by more than the budget ...
    const float g = f.kappa - z2;                                  // ... counted only where z2 > kappa
    in_bits = __builtin_amdgcn_alignbit(in_bits, __float_as_uint(r_in), 31);
    out_bits = __builtin_amdgcn_alignbit(out_bits, __float_as_uint(fmaxf(r_out, g)), 31);
}

// counts[b][slot] = 0 where a model exists (valid byte, or slot%slots_per_sample < nmodels), else -1
__global__ void init_counts_kernel(const uint8_t* __restrict__ valid, const int32_t* __restrict__ nmodels,
                                   int slots_per_sample, int total, int32_t* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    bool ok = true;
    if (valid) ok = valid[i] != 0;
    else if (nmodels) ok = (i % slots_per_sample) < nmodels[i / slots_per_sample];
    counts[i] = ok ? 0 : -1;
}

// Tiers 2 and 3 for one (model, correspondence) pair, model read from global memory.  Deliberately NOT inlined:
// it runs for ~0.2 % of the evaluations, and inlined its ~50 live doubles set the register budget of the whole
// count kernel (104 VGPRs = 4 waves per SIMD instead of 6); as a call, the few registers the hot loop keeps live
// are saved around it instead.
__device__ __attribute__((noinline)) int fund_decide64(const double* __restrict__ model, float x1, float y1, float x2, float y2,
                                                       float thr2, double Q1, double Q2)
{
    double Mm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Mm[k] = model[k];
    const FundFilter g = fund_filter_setup(Mm, (double)thr2, Q1, Q2);
    const int verdict = fund_filter(Mm, g, x1, y1, x2, y2);
    int in = verdict > 0;
    if (verdict == 0) in = fund_err(Mm, x1, y1, x2, y2) <= thr2;
    return in;
}

// A load of an LDS word that the compiler must not cache or hoist (another lane of the wave wrote it).  A volatile
// access through a cast pointer loses the address space and becomes a flat_load with a full vmcnt wait.
template <typename T>
__device__ __forceinline__ T lds_load(T* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ---- count kernel (one template for the three error functions) -----------------------------
enum { kFund = 0, kEss = 1, kPnp = 2 };

struct ScoreArgs {
    const double* models;     // batch x H x MW f64
    const uint8_t* valid;     // batch x H, may be null
    const int32_t* list;      // batch x H compacted slot list, may be null (then thread t scores slot t)
    const int32_t* nstart;    // batch: first list entry to score (may be null = 0)
    const int32_t* nlist;     // batch entries used of list
    const void* pts;          // F: batch x N float4; E: batch x N x 4 f64; PnP: batch x N x 3 f32
    const float2* img;        // PnP only: batch x N
    const int32_t* n_dev;     // per-problem correspondence count (<= N), may be null
    const float* bounds;      // F, MFMA kernel: batch x 4 precomputed coordinate bounds (vo_launch_fund_bounds), may be null
    int32_t* counts;          // batch x H
    int H, N, chunk;
    float thr2;
    double fx, fy, cx, cy;
};

template <int KIND>
__global__ __launch_bounds__(kThreads, KIND == 0 ? VO_SCORE_MINW_F : VO_SCORE_MINW)
void score_kernel(const ScoreArgs a)
{
    constexpr int MW = KIND == kPnp ? 12 : 9;
    __shared__ __attribute__((aligned(16))) double sp[KIND == kFund ? kMaxChunk * 2 : kMaxChunk * 4];
    const int b = blockIdx.z;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const int p0 = blockIdx.y * a.chunk;
    const int first = a.list && a.nstart ? a.nstart[b] : 0;
    const int nslots = a.list ? a.nlist[b] - first : a.H;
    if (p0 >= n || (int)(blockIdx.x * kThreads) >= nslots) return;      // block-uniform
    const int np = min(a.chunk, n - p0);

    __shared__ double sscale[4][kThreads / 64];
    __shared__ float4 sef[KIND == kEss ? kMaxChunk : 1];      // E: the chunk's coordinates rounded to f32
    if (KIND == kFund) {
        const float4* pts = (const float4*)a.pts + (size_t)b * a.N + p0;
        double p1 = 1., p2 = 1.;                   // chunk-wide bounds on max(|x|,|y|,1) of either image
        bool tame = true;                          // every coordinate finite and <= 2^20: the sign-bit filter applies
        for (int i = threadIdx.x; i < np; i += kThreads) {
            const float4 v = pts[i];
            ((float4*)sp)[i] = v;                  // F keeps f32 points: exact in either precision
            const double q1 = fmax(fabs((double)v.x), fabs((double)v.y)), q2 = fmax(fabs((double)v.z), fabs((double)v.w));
            p1 = fmax(p1, q1);                     // a NaN point fails the filter by itself
            p2 = fmax(p2, q2);
            tame = tame && fabsf(v.x) <= 0x1p20f && fabsf(v.y) <= 0x1p20f && fabsf(v.z) <= 0x1p20f && fabsf(v.w) <= 0x1p20f;
        }
        for (int o = 32; o; o >>= 1) {
            const double o1 = __shfl_xor(p1, o), o2 = __shfl_xor(p2, o);
            p1 = fmax(p1, o1);
            p2 = fmax(p2, o2);
        }
        tame = __all(tame);
        if ((threadIdx.x & 63) == 0) {
            sscale[0][threadIdx.x >> 6] = p1; sscale[1][threadIdx.x >> 6] = p2; sscale[2][threadIdx.x >> 6] = tame ? 1. : 0.;
        }
    } else if (KIND == kEss) {
        // f64 coordinates for the exact path, their f32 roundings for the filter, chunk-wide bounds as for F
        const double* pts = (const double*)a.pts + ((size_t)b * a.N + p0) * 4;
        double p1 = 1., p2 = 1.;
        bool tame = true;
        for (int i = threadIdx.x; i < np; i += kThreads) {
            const double x1 = pts[4 * i], y1 = pts[4 * i + 1], x2 = pts[4 * i + 2], y2 = pts[4 * i + 3];
            sp[4 * i] = x1; sp[4 * i + 1] = y1; sp[4 * i + 2] = x2; sp[4 * i + 3] = y2;
            sef[i] = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
            p1 = fmax(p1, fmax(fabs(x1), fabs(y1)));
            p2 = fmax(p2, fmax(fabs(x2), fabs(y2)));
            tame = tame && fabs(x1) <= 0x1p20 && fabs(y1) <= 0x1p20 && fabs(x2) <= 0x1p20 && fabs(y2) <= 0x1p20;
        }
        for (int o = 32; o; o >>= 1) {
            const double o1 = __shfl_xor(p1, o), o2 = __shfl_xor(p2, o);
            p1 = fmax(p1, o1);
            p2 = fmax(p2, o2);
        }
        tame = __all(tame);
        if ((threadIdx.x & 63) == 0) {
            sscale[0][threadIdx.x >> 6] = p1; sscale[1][threadIdx.x >> 6] = p2; sscale[2][threadIdx.x >> 6] = tame ? 1. : 0.;
        }
    } else {
        // 32 B per correspondence, all f32 (the object points are f32 to begin with): (X Y Z ux) (uy iu iv -), with
        // u = (float)(iu - cx), (float)(iv - cy) for the filter tier; plus the chunk-wide bounds the filter needs
        const float* obj = (const float*)a.pts + ((size_t)b * a.N + p0) * 3;
        const float2* img = a.img + (size_t)b * a.N + p0;
        float4* spf = (float4*)sp;
        float pm = 1.f, um = 1.f, im = 1.f;
        bool tame = true;
        for (int i = threadIdx.x; i < np; i += kThreads) {
            const float X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            const float2 uv = img[i];
            const float ux = (float)((double)uv.x - a.cx), uy = (float)((double)uv.y - a.cy);
            spf[2 * i] = make_float4(X, Y, Z, ux);
            spf[2 * i + 1] = make_float4(uy, uv.x, uv.y, 0.f);
            pm = fmaxf(pm, fmaxf(fabsf(X), fmaxf(fabsf(Y), fabsf(Z))));
            um = fmaxf(um, fmaxf(fabsf(ux), fabsf(uy)));
            im = fmaxf(im, fmaxf(fabsf(uv.x), fabsf(uv.y)));
            tame = tame && fabsf(X) <= 0x1p20f && fabsf(Y) <= 0x1p20f && fabsf(Z) <= 0x1p20f && fabsf(uv.x) <= 0x1p20f &&
                   fabsf(uv.y) <= 0x1p20f && fabsf(ux) <= 0x1p20f && fabsf(uy) <= 0x1p20f;
        }
        for (int o = 32; o; o >>= 1) {
            pm = fmaxf(pm, __shfl_xor(pm, o)); um = fmaxf(um, __shfl_xor(um, o)); im = fmaxf(im, __shfl_xor(im, o));
        }
        tame = __all(tame);
        if ((threadIdx.x & 63) == 0) {
            sscale[0][threadIdx.x >> 6] = pm; sscale[1][threadIdx.x >> 6] = um; sscale[2][threadIdx.x >> 6] = tame ? 1. : 0.;
            sscale[3][threadIdx.x >> 6] = im;
        }
    }

    const int t = blockIdx.x * kThreads + threadIdx.x;
    int slot = t;
    bool live = t < nslots;
    if (a.list) slot = live ? a.list[(size_t)b * a.H + first + t] : 0;
    const size_t gslot = (size_t)b * a.H + slot;
    if (!a.list) live = live && (!a.valid || a.valid[gslot]);
    double M[MW];
#pragma unroll
    for (int k = 0; k < MW; ++k) M[k] = live ? a.models[gslot * MW + k] : 0.;
    __syncthreads();
    if (!__any(live)) return;                  // wave past the end of the model list (no barrier follows)
    int cnt = 0;
    if (KIND == kFund) {
        double p1 = sscale[0][0], p2 = sscale[1][0], tame_chunk = sscale[2][0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) {
            const double o1 = sscale[0][w], o2 = sscale[1][w];
            p1 = fmax(p1, o1);
            p2 = fmax(p2, o2);
            tame_chunk = fmin(tame_chunk, sscale[2][w]);
        }
        // The f64 model is only needed to set the filters up and in the rare f64 tiers, which re-read it from
        // global memory (L2) when they run: holding 9 + 5 doubles per lane through the whole kernel costs two waves
        // per SIMD of occupancy, and this kernel needs its waves (5.45 ms at 5 per SIMD, 6.96 ms at 3).
        const FundFilter32 f32 = fund_filter32_setup(M, a.thr2, p1, p2);
        const double* const my_model = a.models + gslot * 9;
        const float4* spf = (const float4*)sp;
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        // owner-lane tier 2 over a bit string (bit nb-1-k = correspondence first + k): the paths that are not pooled
        const auto own_tier2 = [&](unsigned long long bits, int first, int nb) {
            while (__any(bits != 0)) {
                if (bits) {
                    const int pos = 63 - __clzll((long long)bits);
                    bits &= ~(1ull << pos);
                    const float4 v = spf[first + nb - 1 - pos];
                    cnt += fund_decide64(my_model, v.x, v.y, v.z, v.w, a.thr2, p1, p2);
                }
            }
        };
        // sign-bit forms of tier 1 for full blocks of tame chunks (see fund_filter32_signs); a lane whose model or
        // threshold falls outside their domain takes every correspondence to tier 2 instead (force)
        const bool fast = tame_chunk != 0.;                                         // workgroup-uniform
        const uint32_t force = (__builtin_isinf(f32.eta2) || !(a.thr2 >= 0x1p-60f && a.thr2 <= 0x1p40f)) ? 0xffffffffu : 0u;
        // Full blocks of 64 correspondences of a tame chunk.
        //   tier 1a: the one-sided sure-outlier test for every correspondence (14 instructions, sign bits);
        //   tier 1b: the complete f32 filter for the ~2.5 % that survive it.  Survivors are very unevenly spread over
        //            the models (a lane's own list is 10.7 entries long at the worst lane of a wave, 1.6 on average), so
        //            the wave pools them: every lane appends its (model lane, correspondence) pairs to a list in LDS,
        //            then lane l takes entries l, l + 64, ... whatever model they belong to, fetching that model's
        //            filter constants from the owner lane's registers (ds_bpermute), and adds the verdict to the
        //            owner's LDS counter or queues the pair for tier 2;
        //   tier 2:  the f64 filter and the exact evaluation for what even that cannot decide (0.23 %): queued per
        //            wave and worked off 64 pairs at a time, one per lane, the model re-read from global memory.
        // A block with more survivors than the list holds is taken by the owners themselves, one per lane per trip.
        int i0 = 0;
        if (fast) {
            // kListCap: survivors one block may add to the tier-1b queue (127 can be waiting); the tier-2 queue takes
            // at most 128 per flush1 on top of the 63 it may keep
            constexpr int kNE = VO_SCORE_NE, kBatch = 64 * kNE;     // tier-1b pairs per lane per flush
            constexpr int kWaves = kThreads / 64, kListCap = 384, kQueueCap = 64 + kBatch;
            __shared__ uint16_t wlist[kWaves][kBatch + kListCap]; // tier-1b queue: (model lane << 9) | correspondence of the chunk
            __shared__ uint16_t wq2[kWaves][kQueueCap];       // tier-2 queue, same entries
            __shared__ int wcnt[kWaves][64];
            __shared__ uint32_t wslot[kWaves][64];            // model slot of every lane, for the tier-2 re-read
            __shared__ uint32_t wtot[kWaves], wq2n[kWaves];
#if VO_SCORE_CONST_LDS
            __shared__ __attribute__((aligned(16))) float wflt[kWaves][64][12];   // F[9], eta2, c of every lane's model
            {
                float4* o = (float4*)wflt[w][lane];
                o[0] = make_float4(f32.F[0], f32.F[1], f32.F[2], f32.F[3]);
                o[1] = make_float4(f32.F[4], f32.F[5], f32.F[6], f32.F[7]);
                o[2] = make_float4(f32.F[8], f32.eta2, f32.c, 0.f);
            }
#endif
            wcnt[w][lane] = 0;
            wslot[w][lane] = (uint32_t)gslot;
            if (lane == 0) { wtot[w] = 0; wq2n[w] = 0; }
            const auto flush2 = [&](uint32_t take, uint32_t base) {
                if ((uint32_t)lane < take) {
                    const uint32_t ent = wq2[w][base + lane];
                    const int m = ent >> 9, i = ent & 511;
                    const float4 v = spf[i];
                    if (fund_decide64(a.models + (size_t)wslot[w][m] * 9, v.x, v.y, v.z, v.w, a.thr2, p1, p2)) atomicAdd(&wcnt[w][m], 1);
#ifdef VO_SCORE_STATS
                    atomicAdd(&g_stats[2], 1ull);
#endif
                }
#ifdef VO_SCORE_STATS
                if (lane == 0) atomicAdd(&g_stats[6], 1ull);
#endif
            };
            // tier 1b for `take` (<= 128) queued pairs starting at `base`: lane l takes entries l and l + 64 -- two
            // independent dependency chains per lane, the loop is bound by LDS round trips, not by issue
            const auto flush1 = [&](uint32_t take, uint32_t base) {
                bool has[kNE];
                uint32_t ent[kNE];
#pragma unroll
                for (int j = 0; j < kNE; ++j) {
                    has[j] = (uint32_t)(lane + 64 * j) < take;
                    ent[j] = has[j] ? wlist[w][base + lane + 64 * j] : 0u;
                }
                int verdict[kNE];
#pragma unroll
                for (int j = 0; j < kNE; ++j) {
                    const int m = ent[j] >> 9, i = ent[j] & 511;
                    FundFilter32 g;
#if VO_SCORE_CONST_LDS
                    {
                        const float4* fm = (const float4*)wflt[w][m];
                        const float4 A = fm[0], B = fm[1], C = fm[2];
                        g.F[0] = A.x; g.F[1] = A.y; g.F[2] = A.z; g.F[3] = A.w; g.F[4] = B.x; g.F[5] = B.y; g.F[6] = B.z; g.F[7] = B.w;
                        g.F[8] = C.x; g.eta2 = C.y; g.c = C.z;
                    }
#else
#pragma unroll
                    for (int q = 0; q < 9; ++q) g.F[q] = __shfl(f32.F[q], m);
                    g.eta2 = __shfl(f32.eta2, m); g.c = __shfl(f32.c, m);
#endif
                    g.thi = f32.thi; g.tlo = f32.tlo;                                  // functions of the threshold alone
                    const float4 v = spf[i];
                    verdict[j] = fund_filter32(g, v.x, v.y, v.z, v.w);
                }
#pragma unroll
                for (int j = 0; j < kNE; ++j) {
                    if (has[j]) {
                        const int m = ent[j] >> 9;
                        if (verdict[j] > 0) atomicAdd(&wcnt[w][m], 1);
                        else if (verdict[j] == 0) wq2[w][atomicAdd(&wq2n[w], 1u)] = (uint16_t)ent[j];
                    }
                }
#ifdef VO_SCORE_STATS
                if (lane == 0) atomicAdd(&g_stats[1], 1ull);
#endif
            };
            // the wave's lanes talk through LDS here: LDS instructions of one wave execute in order, the fences only
            // keep the compiler from moving them
            const auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
            const auto drain2 = [&](uint32_t keep) {        // work the tier-2 queue off while it holds more than `keep` entries
                uint32_t queued = __builtin_amdgcn_readfirstlane(lds_load(&wq2n[w]));
                if (queued > keep) {
                    do {
                        const uint32_t take = min(queued, 64u);
                        queued -= take;
                        flush2(take, queued);
                    } while (queued > keep);
                    wave_sync();
                    if (lane == 0) wq2n[w] = queued;
                    wave_sync();
                }
            };
            wave_sync();
            uint32_t q1 = 0;                                 // wave-uniform: entries waiting in the tier-1b queue
            for (; i0 + 64 <= np; i0 += 64) {
                uint32_t o0 = 0, o1 = 0;
#pragma unroll VO_SCORE_G
                for (int k = 0; k < 32; ++k) {
                    const float4 v = spf[i0 + k];
                    fund_filter32_outlier_sign(f32, v.x, v.y, v.z, v.w, o0);
                }
#pragma unroll VO_SCORE_G
                for (int k = 0; k < 32; ++k) {
                    const float4 v = spf[i0 + 32 + k];
                    fund_filter32_outlier_sign(f32, v.x, v.y, v.z, v.w, o1);
                }
                unsigned long long cand = ~(((unsigned long long)o0 << 32) | o1);      // bit 63 - k = correspondence i0 + k
                unsigned long long und = 0;
                if (force) { cand = 0; und = ~0ull; }          // model or threshold outside the f32 filter's domain: all to tier 2
                if (!live) { cand = 0; und = 0; }
                uint32_t at = atomicAdd(&wtot[w], (uint32_t)__popcll(cand));           // this lane's place in the queue
                wave_sync();
                const uint32_t q1new = __builtin_amdgcn_readfirstlane(lds_load(&wtot[w]));
#ifdef VO_SCORE_STATS
                atomicAdd(&g_stats[0], (unsigned long long)__popcll(cand));
                if (lane == 0) atomicAdd(&g_stats[3], 1ull);
#endif
                if (q1new - q1 <= (uint32_t)kListCap) {
                    const int idx0 = (lane << 9) | (i0 + 63);
                    while (cand) {
                        const int bit = __ffsll((long long)cand) - 1;
                        cand &= cand - 1;
                        wlist[w][at++] = (uint16_t)(idx0 - bit);                        // (model lane << 9) | correspondence of the chunk
                    }
                    q1 = q1new;
                    wave_sync();
                    if (q1 >= (uint32_t)kBatch) {
                        do {
                            q1 -= kBatch;
                            flush1(kBatch, q1);
                            wave_sync();
                            drain2(63);                     // at most 63 + 128 entries ever wait in the tier-2 queue
                        } while (q1 >= (uint32_t)kBatch);
                        if (lane == 0) wtot[w] = q1;
                        wave_sync();
                    }
                } else {
                    // more survivors in one block than the queue has room for: the owners take them, one per lane per trip
                    wave_sync();
                    if (lane == 0) wtot[w] = q1;
                    wave_sync();
                    while (__any(cand != 0)) {
                        if (cand) {
                            const int pos = 63 - __clzll((long long)cand);
                            cand &= ~(1ull << pos);
                            const float4 v = spf[i0 + 63 - pos];
                            const int verdict = fund_filter32(f32, v.x, v.y, v.z, v.w);
                            cnt += verdict > 0;
                            if (verdict == 0) und |= 1ull << pos;
                        }
                    }
                }
#ifdef VO_SCORE_STATS
                atomicAdd(&g_stats[2], (unsigned long long)__popcll(und));
                if (lane == 0 && __any(und != 0)) atomicAdd(&g_stats[4], 1ull);
#endif
                own_tier2(und, i0, 64);
            }
            wave_sync();
            if (q1) flush1(q1, 0);
            wave_sync();
            drain2(0);
            wave_sync();
            cnt += lds_load(&wcnt[w][lane]);
        }
        // ragged tail of the chunk, and chunks that are not tame: blocks of 32, tier 2 by the owner lane
        for (; i0 < np; i0 += 32) {
            const int nb = min(32, np - i0);
            uint32_t bits = 0;
            auto tier1 = [&](int k) {
                const float4 v = spf[i0 + k];
                const int verdict = fund_filter32(f32, v.x, v.y, v.z, v.w);
                cnt += verdict > 0;
                bits = (bits << 1) | (uint32_t)(verdict == 0);
            };
            if (nb == 32 && fast) {
                uint32_t in_bits = 0, out_bits = 0;
#pragma unroll VO_SCORE_G
                for (int k = 0; k < 32; ++k) {
                    const float4 v = spf[i0 + k];
                    fund_filter32_signs(f32, v.x, v.y, v.z, v.w, in_bits, out_bits);
                }
                bits = ~(in_bits | out_bits) | force;
                cnt += __popc(in_bits & ~force);
            } else if (nb == 32) {
#pragma unroll VO_SCORE_G
                for (int k = 0; k < 32; ++k) tier1(k);
            } else {
#pragma unroll 1
                for (int k = 0; k < nb; ++k) tier1(k);
            }
            if (!live) bits = 0;
            own_tier2(bits, i0, nb);
        }
    } else if (KIND == kPnp) {
        double pm = sscale[0][0], um = sscale[1][0], tame_chunk = sscale[2][0], im = sscale[3][0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) {
            pm = fmax(pm, sscale[0][w]); um = fmax(um, sscale[1][w]); im = fmax(im, sscale[3][w]);
            tame_chunk = fmin(tame_chunk, sscale[2][w]);
        }
        const float4* spf = (const float4*)sp;
        auto exact = [&](int i) -> int {
            const float4 p = spf[2 * i], q = spf[2 * i + 1];
            return pnp_err(M, a.fx, a.fy, a.cx, a.cy, p.x, p.y, p.z, q.y, q.z) <= a.thr2;
        };
        const PnpFilter32 pf = pnp_filter32_setup(M, a.fx, a.fy, a.thr2, pm, um, im);
        const bool fast = tame_chunk != 0.;                                         // workgroup-uniform
        const uint32_t force = pf.usable ? 0u : 0xffffffffu;
        int i0 = 0;
        if (fast) {
            for (; i0 + 32 <= np; i0 += 32) {
                uint32_t in_bits = 0, out_bits = 0;
#pragma unroll VO_SCORE_GP
                for (int k = 0; k < 32; ++k) {
                    const float4 p = spf[2 * (i0 + k)];
                    const float uy = ((const float*)sp)[8 * (i0 + k) + 4];
                    pnp_filter32_signs(pf, p.x, p.y, p.z, p.w, uy, in_bits, out_bits);
                }
                uint32_t bits = ~(in_bits | out_bits) | force;
                cnt += __popc(in_bits & ~force);
                if (!live) bits = 0;
                // tier 2: the exact evaluation for what the filter could not decide, one entry per lane per trip
                while (__any(bits != 0)) {
                    if (bits) {
                        const int pos = 31 - __clz(bits);
                        bits &= ~(1u << pos);
                        cnt += exact(i0 + 31 - pos);
                    }
                }
            }
        }
        for (; i0 < np; ++i0) cnt += exact(i0);
    } else {
        double q1 = sscale[0][0], q2 = sscale[1][0], tame_chunk = sscale[2][0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) {
            q1 = fmax(q1, sscale[0][w]); q2 = fmax(q2, sscale[1][w]);
            tame_chunk = fmin(tame_chunk, sscale[2][w]);
        }
        auto exact = [&](int i) -> int {
            const double2 u = *(const double2*)(sp + 4 * i);
            const double2 v = *(const double2*)(sp + 4 * i + 2);
            return sampson_err(M, u.x, u.y, v.x, v.y) <= a.thr2;
        };
        const FundFilter32 ef = filter32_setup<true>(M, a.thr2, q1, q2);
        const bool fast = tame_chunk != 0.;                                         // workgroup-uniform
        const uint32_t force = (__builtin_isinf(ef.eta2) || !(a.thr2 >= 0x1p-60f && a.thr2 <= 0x1p40f)) ? 0xffffffffu : 0u;
        int i0 = 0;
        if (fast) {
            for (; i0 + 32 <= np; i0 += 32) {
                uint32_t in_bits = 0, out_bits = 0;
#pragma unroll VO_SCORE_GP
                for (int k = 0; k < 32; ++k) {
                    const float4 v = sef[i0 + k];
                    ess_filter32_signs(ef, v.x, v.y, v.z, v.w, in_bits, out_bits);
                }
                uint32_t bits = ~(in_bits | out_bits) | force;
                cnt += __popc(in_bits & ~force);
                if (!live) bits = 0;
                // tier 2: the exact evaluation for what the filter could not decide, one entry per lane per trip
                while (__any(bits != 0)) {
                    if (bits) {
                        const int pos = 31 - __clz(bits);
                        bits &= ~(1u << pos);
                        cnt += exact(i0 + 31 - pos);
                    }
                }
            }
        }
        for (; i0 < np; ++i0) cnt += exact(i0);
    }
    if (live && cnt) atomicAdd(a.counts + gslot, cnt);
}

#ifndef VO_SCORE_F_MFMA
#define VO_SCORE_F_MFMA 1             // 1: the F count kernel runs its first tier on the matrix cores (vo_score_fmfma.h)
#endif
#ifndef VO_SCORE_MF_TARGET
#define VO_SCORE_MF_TARGET 4096       // workgroups wanted (4 rounds of the 1024 resident ones) before one walks several chunks; cfg5: 1024 -> 4096 = 2.40 -> 1.74 ms
#endif
#include "vo_score_fmfma.h"

// ---- mask kernels (one model, thread per correspondence) ----------------------------------

__global__ void mask_fund_kernel(const double* __restrict__ model, const float4* __restrict__ pts, int N,
                                 float thr2, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = model[k];
    const float4 v = pts[i];
    mask[i] = fund_err(F, v.x, v.y, v.z, v.w) <= thr2;
}

__global__ void mask_ess_kernel(const double* __restrict__ model, const double* __restrict__ pts, int N,
                                float thr2, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = model[k];
    const double* p = pts + (size_t)i * 4;
    mask[i] = sampson_err(E, p[0], p[1], p[2], p[3]) <= thr2;
}

__global__ void mask_pnp_kernel(const double* __restrict__ model, const float* __restrict__ obj,
                                const float2* __restrict__ img, int N, double fx, double fy, double cx,
                                double cy, float thr2, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = model[k];
    const float* o = obj + (size_t)i * 3;
    const float2 uv = img[i];
    mask[i] = pnp_err(M, fx, fy, cx, cy, o[0], o[1], o[2], uv.x, uv.y) <= thr2;
}

// Mask of the model the replay chose, for a batch of problems: result[b] = {best_iter, best_model, ...}
// addresses the model table directly, so no host round trip sits between replay and mask.
struct MaskArgs {
    const double* models;      // batch x slots_per_problem x MW
    const int32_t* result;     // batch x 4
    const void* pts;
    const float2* img;
    const int32_t* n_dev;
    uint8_t* mask;             // batch x N
    double* best_model;        // batch x MW, may be null
    const double* aux;         // batch x (slots_per_problem) x aux_w, may be null (P3P rotation vectors)
    double* best_aux;          // batch x aux_w
    int slots_per_problem, slots_per_sample, aux_w, N;
    int direct;                // 1: models is batch x MW, one model per problem (the winner itself)
    float thr2;
    double fx, fy, cx, cy;
};

// best_model[b] / best_aux[b] = the winner's model where this rank solved it (iteration in [own_lo, own_hi)),
// zeros elsewhere: summed over the ranks as integers, every rank ends up with the winner's bits.
__global__ __launch_bounds__(64)
void select_owned_kernel(const double* __restrict__ models, const double* __restrict__ aux, const int32_t* __restrict__ result,
                         int slots_per_problem, int slots_per_sample, int mw, int aux_w, int own_lo, int own_hi, int batch,
                         double* __restrict__ best_model, double* __restrict__ best_aux)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int best_iter = result[4 * b], best_k = result[4 * b + 1];
    const bool own = best_iter >= own_lo && best_iter < own_hi && best_k >= 0;
    const size_t slot = (size_t)b * slots_per_problem + (own ? best_iter * slots_per_sample + best_k : 0);
    for (int k = 0; k < mw; ++k) best_model[(size_t)b * mw + k] = own ? models[slot * mw + k] : 0.;
    if (aux && best_aux)
        for (int k = 0; k < aux_w; ++k) best_aux[(size_t)b * aux_w + k] = own ? aux[slot * aux_w + k] : 0.;
}

template <int KIND>
__global__ __launch_bounds__(256)
void mask_best_kernel(const MaskArgs a)
{
    constexpr int MW = KIND == kPnp ? 12 : 9;
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const int best_iter = a.result[4 * b], best_k = a.result[4 * b + 1];
    const bool have = best_iter >= 0 && best_k >= 0;
    const size_t slot = a.direct ? (size_t)b
                                 : (size_t)b * a.slots_per_problem + (have ? best_iter * a.slots_per_sample + best_k : 0);
    double M[MW];
#pragma unroll
    for (int k = 0; k < MW; ++k) M[k] = have ? a.models[slot * MW + k] : 0.;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.best_model)
            for (int k = 0; k < MW; ++k) a.best_model[(size_t)b * MW + k] = M[k];
        if (a.aux && a.best_aux)
            for (int k = 0; k < a.aux_w; ++k) a.best_aux[(size_t)b * a.aux_w + k] = have ? a.aux[slot * a.aux_w + k] : 0.;
    }
    if (i >= a.N) return;
    uint8_t m = 0;
    if (have && i < n) {
        const size_t gi = (size_t)b * a.N + i;
        if (KIND == kFund) {
            const float4 v = ((const float4*)a.pts)[gi];
            m = fund_err(M, v.x, v.y, v.z, v.w) <= a.thr2;
        } else if (KIND == kEss) {
            const double* p = (const double*)a.pts + gi * 4;
            m = sampson_err(M, p[0], p[1], p[2], p[3]) <= a.thr2;
        } else {
            const float* o = (const float*)a.pts + gi * 3;
            const float2 uv = a.img[gi];
            m = pnp_err(M, a.fx, a.fy, a.cx, a.cy, o[0], o[1], o[2], uv.x, uv.y) <= a.thr2;
        }
    }
    a.mask[(size_t)b * a.N + i] = m;
}

// LMedS ranking (OpenCV's LMeDSPointSetRegistrator, used by findFundamentalMat for 8..14 points): per
// model the element of rank N/2 among its f32 errors ordered as their int reinterpretation (what
// std::nth_element over errf.ptr<int>() yields).  Thread per model, N <= kLmedsMaxPoints.
constexpr int kLmedsMaxPoints = 64;

__global__ __launch_bounds__(64)
void lmeds_median_fund_kernel(const double* __restrict__ models, const uint8_t* __restrict__ valid, int H,
                              const float4* __restrict__ pts, int N, float* __restrict__ medians)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    if (valid && !valid[h]) { medians[h] = __builtin_huge_valf(); return; }
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = models[(size_t)h * 9 + k];
    int key[kLmedsMaxPoints];
    for (int i = 0; i < N; ++i) {                       // insertion sort of the int keys
        const float4 v = pts[i];
        const int e = __float_as_int(fund_err(F, v.x, v.y, v.z, v.w));
        int j = i;
        for (; j > 0 && key[j - 1] > e; --j) key[j] = key[j - 1];
        key[j] = e;
    }
    medians[h] = __int_as_float(key[N / 2]);
}

// ---- frame pairs with 7..14 matches inside a batch (cv2.findFundamentalMat's small-input branches)
// n_lmeds[b] = n where OpenCV runs its LMedS registrator (8..14 points), else 0 (getSubset then produces no sample);
// n_small[b] = n for 7..14, else 0; a pair with exactly 7 matches gets the one sample 0..6 (OpenCV solves the seven
// points as they stand)
__global__ void small_prepare_kernel(const int32_t* __restrict__ n_match, int batch, int H, int32_t* __restrict__ n_lmeds,
                                     int32_t* __restrict__ n_small, int32_t* __restrict__ samples)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int n = n_match[b];
    n_lmeds[b] = n >= 8 && n <= 14 ? n : 0;
    n_small[b] = n >= 7 && n <= 14 ? n : 0;
}

__global__ void small_identity_kernel(const int32_t* __restrict__ n_match, int batch, int H, int32_t* __restrict__ samples)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch || n_match[b] != 7) return;
    for (int i = 0; i < 7; ++i) samples[(size_t)b * H * 7 + i] = i;
}

// One wave per pair.  8..14 matches: OpenCV's LMeDSPointSetRegistrator -- every model ranked by the median of its
// errors (element N/2 of the errors ordered as their int reinterpretation), the first strictly smallest median
// wins, inliers lie within sigma = 2.5 * 1.4826 * (1 + 5/(n-7)) * sqrt(median) (at least 0.001), fewer than 7
// inliers = no model.  Exactly 7 matches: all solutions of the one sample are returned (models7 / nmodels7), the
// mask is all ones and F is the first solution.  Pairs outside 7..14 are left untouched.
__global__ __launch_bounds__(64)
void small_finish_kernel(const double* __restrict__ models, const int32_t* __restrict__ nmodels, int H,
                         const float4* __restrict__ pts, int N, const int32_t* __restrict__ n_small, int32_t* __restrict__ result,
                         double* __restrict__ Fout, uint8_t* __restrict__ mask, double* __restrict__ models7,
                         int32_t* __restrict__ nmodels7)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = n_small[b];
    if (n == 0) return;
    pts += (size_t)b * N;
    models += (size_t)b * H * 27;
    nmodels += (size_t)b * H;
    mask += (size_t)b * N;
    if (n == 7) {
        const int k = nmodels[0];
        if (lane < 27) models7[(size_t)b * 27 + lane] = lane < 9 * k ? models[lane] : 0.;
        if (lane < 9) Fout[(size_t)b * 9 + lane] = k > 0 ? models[lane] : 0.;
        for (int i = lane; i < N; i += 64) mask[i] = i < 7;
        if (lane == 0) {
            nmodels7[b] = k;
            int32_t* r = result + 4 * (size_t)b;
            r[0] = k > 0 ? 0 : -1; r[1] = k > 0 ? 0 : -1; r[2] = k > 0 ? 7 : 0; r[3] = 0;
        }
        return;
    }
    // slot = 3 * iteration + model, visited in that order, 64 per trip
    float best_med = __builtin_huge_valf();
    int best_slot = -1;
    for (int base = 0; base < 3 * H; base += 64) {
        const int slot = base + lane, it = slot / 3, k = slot - 3 * it;
        float med = __builtin_huge_valf();
        if (slot < 3 * H && k < nmodels[it]) {
            double F[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) F[q] = models[(size_t)slot * 9 + q];
            int key[16];
            for (int i = 0; i < n; ++i) {                       // insertion sort of the int keys (n <= 14)
                const float4 v = pts[i];
                const int e = __float_as_int(fund_err(F, v.x, v.y, v.z, v.w));
                int j = i;
                for (; j > 0 && key[j - 1] > e; --j) key[j] = key[j - 1];
                key[j] = e;
            }
            med = __int_as_float(key[n / 2]);
        }
        // "median < minMedian" in slot order: the smallest median of the trip at its lowest slot, then strict <
        float m = med;
        int sidx = med < __builtin_huge_valf() ? slot : 0x7fffffff;      // NaN compares false: never a candidate
        if (!(med < __builtin_huge_valf())) m = __builtin_huge_valf();
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const float om = __shfl_xor(m, o);
            const int os = __shfl_xor(sidx, o);
            if (om < m || (om == m && os < sidx)) { m = om; sidx = os; }
        }
        if (m < best_med) { best_med = m; best_slot = sidx; }
    }
    const int niters = H;
    if (best_slot < 0) {
        for (int i = lane; i < N; i += 64) mask[i] = 0;
        if (lane < 9) Fout[(size_t)b * 9 + lane] = 0.;
        if (lane == 0) { int32_t* r = result + 4 * (size_t)b; r[0] = -1; r[1] = -1; r[2] = 0; r[3] = niters; }
        return;
    }
    double F[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) F[q] = models[(size_t)best_slot * 9 + q];
    double sigma = 2.5 * 1.4826 * (1. + 5. / (double)(n - 7)) * sqrt((double)best_med);
    sigma = sigma > 0.001 ? sigma : 0.001;
    const float thr2 = (float)(sigma * sigma);
    int good = 0;
    for (int i = lane; i < N; i += 64) {
        uint8_t mm = 0;
        if (i < n) {
            const float4 v = pts[i];
            mm = fund_err(F, v.x, v.y, v.z, v.w) <= thr2;
        }
        mask[i] = mm;
        good += mm;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) good += __shfl_xor(good, o);
    if (lane < 9) Fout[(size_t)b * 9 + lane] = good >= 7 ? F[lane] : 0.;
    if (lane == 0) {
        int32_t* r = result + 4 * (size_t)b;
        r[0] = good >= 7 ? best_slot / 3 : -1; r[1] = good >= 7 ? best_slot % 3 : -1; r[2] = good >= 7 ? good : 0; r[3] = niters;
    }
}

int pick_chunk(int H, int N, int batch)
{
    const long mblocks = (long)vo_ceil_div(H, kThreads) * batch;
    const long target = 2048;                     // ~8 workgroups of 4 waves per CU
    long per = (target + mblocks - 1) / mblocks;  // chunks wanted along N
    if (per < 1) per = 1;
    int chunk = vo_ceil_div(N, (int)per);
    chunk = vo_ceil_div(chunk, 32) * 32;
    if (chunk < 32) chunk = 32;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    return chunk;
}

}  // namespace

// shared with vo_ransac.hip (the fused batched pipelines)
int vo_launch_init_counts(const uint8_t* valid, const int32_t* nmodels, int slots_per_sample, long total,
                          int32_t* counts, hipStream_t s)
{
    VO_CHECK_ARG(total <= 2147483647L, VO_ESIZE, "too many model slots");
    if (total == 0) return VO_OK;
    hipLaunchKernelGGL(init_counts_kernel, dim3(vo_ceil_div((int)total, 256)), dim3(256), 0, s, valid, nmodels,
                       slots_per_sample, (int)total, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

namespace {
// one workgroup per problem: what every workgroup of score_fund_mfma_kernel would otherwise find for itself
__global__ __launch_bounds__(256)
void fund_bounds_kernel(const float4* __restrict__ pts, const int32_t* __restrict__ n_dev, int N, float* __restrict__ bounds)
{
    __shared__ float sq[3][4];
    const int b = blockIdx.x, n = n_dev ? min(n_dev[b], N) : N;
    pts += (size_t)b * N;
    float q1 = 1.f, q2 = 1.f;
    bool tame = true;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float4 v = pts[i];
        q1 = fmaxf(q1, fmaxf(fabsf(v.x), fabsf(v.y)));
        q2 = fmaxf(q2, fmaxf(fabsf(v.z), fabsf(v.w)));
        tame = tame && fabsf(v.x) <= 0x1p20f && fabsf(v.y) <= 0x1p20f && fabsf(v.z) <= 0x1p20f && fabsf(v.w) <= 0x1p20f;
    }
    for (int o = 32; o; o >>= 1) {
        q1 = fmaxf(q1, __shfl_xor(q1, o));
        q2 = fmaxf(q2, __shfl_xor(q2, o));
    }
    tame = __all(tame);
    if ((threadIdx.x & 63) == 0) { sq[0][threadIdx.x >> 6] = q1; sq[1][threadIdx.x >> 6] = q2; sq[2][threadIdx.x >> 6] = tame ? 1.f : 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 o;
        o.x = fmaxf(fmaxf(sq[0][0], sq[0][1]), fmaxf(sq[0][2], sq[0][3]));
        o.y = fmaxf(fmaxf(sq[1][0], sq[1][1]), fmaxf(sq[1][2], sq[1][3]));
        o.z = fminf(fminf(sq[2][0], sq[2][1]), fminf(sq[2][2], sq[2][3]));
        o.w = 0.f;
        *(float4*)(bounds + 4 * (size_t)b) = o;
    }
}
}  // namespace

int vo_launch_fund_bounds(const void* pts, const int32_t* n_dev, int N, int batch, float* bounds, hipStream_t s)
{
    if (batch == 0 || N == 0) return VO_OK;
    hipLaunchKernelGGL(fund_bounds_kernel, dim3(batch), dim3(256), 0, s, (const float4*)pts, n_dev, N, bounds);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_score(int kind, const double* models, const uint8_t* valid, const int32_t* list, const int32_t* nstart,
                    const int32_t* nlist, int H, const void* pts, const float* img, const int32_t* n_dev, int N, int batch,
                    const double* K4, float thr2, int32_t* counts, int max_slots, hipStream_t s, const float* fund_bounds)
{
    // H = model slots per problem (array stride); max_slots <= H bounds how many this launch can score
    if (H == 0 || N == 0 || batch == 0) return VO_OK;
    ScoreArgs a;
    a.models = models; a.valid = valid; a.list = list; a.nstart = nstart; a.nlist = nlist; a.pts = pts; a.img = (const float2*)img;
    a.n_dev = n_dev; a.counts = counts; a.H = H; a.N = N; a.thr2 = thr2; a.bounds = fund_bounds;
    if (max_slots <= 0 || max_slots > H) max_slots = H;
    if (kind == kFund && VO_SCORE_F_MFMA) {
        // chunks of 512 correspondences; a workgroup walks every gy-th chunk with its models' operands in registers
        const int nchunks = vo_ceil_div(N, kMfChunk);
        const long mblocks = (long)vo_ceil_div(max_slots, kThreads) * batch;
        long gy = (VO_SCORE_MF_TARGET + mblocks - 1) / mblocks;
        if (gy > nchunks) gy = nchunks;
        if (gy < 1) gy = 1;
        VO_CHECK_ARG(batch <= 65535, VO_ESIZE, "score: batch=%d too large", batch);
        a.chunk = kMfChunk;
        a.fx = a.fy = a.cx = a.cy = 0;
        const dim3 grid(vo_ceil_div(max_slots, kThreads), (unsigned)gy, batch);
        hipLaunchKernelGGL(score_fund_mfma_kernel, grid, dim3(kThreads), 0, s, a, (int)gy);
        VO_LAUNCH_CHECK();
        return VO_OK;
    }
    a.chunk = pick_chunk(max_slots, N, batch);
    a.fx = K4 ? K4[0] : 0; a.fy = K4 ? K4[1] : 0; a.cx = K4 ? K4[2] : 0; a.cy = K4 ? K4[3] : 0;
    VO_CHECK_ARG(vo_ceil_div(N, a.chunk) <= 65535 && batch <= 65535, VO_ESIZE, "score: N=%d or batch=%d too large", N, batch);
    const dim3 grid(vo_ceil_div(max_slots, kThreads), vo_ceil_div(N, a.chunk), batch);
    if (kind == kFund) hipLaunchKernelGGL(score_kernel<kFund>, grid, dim3(kThreads), 0, s, a);
    else if (kind == kEss) hipLaunchKernelGGL(score_kernel<kEss>, grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(score_kernel<kPnp>, grid, dim3(kThreads), 0, s, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

namespace {
// exclusive prefix sums of the per-problem contender counts (one counter per kNclStride ints): pre[b] = ncl[0] + ... + ncl[b-1],
// pre[batch] = their number
__global__ __launch_bounds__(1024)
void contender_prefix_kernel(const int32_t* __restrict__ ncl, int batch, int32_t* __restrict__ pre)
{
    __shared__ int part[1024];
    const int per = (batch + 1023) / 1024, lo = threadIdx.x * per, hi = min(batch, lo + per);
    int sum = 0;
    for (int b = lo; b < hi; ++b) sum += ncl[(size_t)b * kNclStride];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int b = lo; b < hi; ++b) { pre[b] = run; run += ncl[(size_t)b * kNclStride]; }
    if (threadIdx.x == 1023) pre[batch] = part[1023];
}

// Exact counts of the contenders of a pruned stage: they are few (a hundred or so per problem, very unevenly spread over
// the problems) and they are the GOOD hypotheses, dense in survivors -- the opposite of what score_fund_mfma_kernel's queues
// are built for -- so each is simply walked by a whole workgroup, thread per correspondence: the complete f32 filter, the
// f64 filter and the definition where it cannot tell.  The contenders of all problems form one index space (prefix sums
// of the per-problem list lengths): workgroup g takes contenders g, g + gridDim.x, ... of the whole batch, whichever
// problem they belong to.  counts[slot] holds the bound on entry (screen_finalize_kernel) and the count on exit.
__global__ __launch_bounds__(256)
void count_contenders_kernel(const double* __restrict__ models, const int32_t* __restrict__ clist, const int32_t* __restrict__ pre,
                             int batch, int Hs, const float4* __restrict__ pts_all, const int32_t* __restrict__ n_dev, int N,
                             float thr2, const float* __restrict__ bounds, int32_t* __restrict__ counts)
{
    constexpr int kUndCap = 1024;
    __shared__ int wsum[4];
    __shared__ FundFilter32 sf32;
    __shared__ FundFilter sg64;
    __shared__ int und[kUndCap];
    __shared__ int nund;
    const int total = pre[batch];
    for (int g = blockIdx.x; g < total; g += gridDim.x) {   // (everything that decides about g is workgroup-uniform)
        int lo = 0, hi = batch;                              // the problem: the last b with pre[b] <= g
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= g) lo = mid; else hi = mid;
        }
        const int b = lo;
        const int n = n_dev ? min(n_dev[b], N) : N;
        const float4* const pts = pts_all + (size_t)b * N;
        const float4 pb = *(const float4*)(bounds + 4 * (size_t)b);
        const double p1 = (double)pb.x, p2 = (double)pb.y;
        const size_t gs = (size_t)b * Hs + clist[(size_t)b * Hs + (g - pre[b])];
        const double* const model = models + gs * 9;
        double M[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = model[k];
        // fund_decide64's tiers with the f64 filter constants set up once per contender instead of once per call: a good
        // model has many residuals next to the threshold, some lane of every wave needs them in nearly every trip.  The two
        // set-ups are a few thousand f64 instructions: one wave does them for the workgroup (done by all four they were two
        // thirds of this kernel's instructions)
        __syncthreads();                                     // the previous contender's constants and sums have been read
        if (threadIdx.x < 64) {
            const FundFilter32 f = fund_filter32_setup(M, thr2, p1, p2);
            const FundFilter g = fund_filter_setup(M, (double)thr2, p1, p2);
            if (threadIdx.x == 0) { sf32 = f; sg64 = g; nund = 0; }
        }
        __syncthreads();
        const FundFilter32 f32 = sf32;
        const FundFilter g64 = sg64;
        int cnt = 0;
        for (int i0 = 0; i0 < n; i0 += 4 * 256) {           // four loads in flight per thread (prefetching the next four: slower)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + threadIdx.x;
                v[u] = i < n ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + (int)threadIdx.x;
                int verdict = i < n ? fund_filter32(f32, v[u].x, v[u].y, v[u].z, v[u].w) : -1;
                if (verdict == 0) {
                    // a good model has a residual next to the threshold in most trips of most waves, and one undecided lane
                    // makes its whole wave walk the f64 tiers: pool them over the workgroup and decide them together below
                    const int pos = atomicAdd(&nund, 1);
                    if (pos < kUndCap) {
                        und[pos] = i;
                    } else {                                 // (list full: decide here)
                        verdict = fund_filter(M, g64, v[u].x, v[u].y, v[u].z, v[u].w);
                        if (verdict == 0) verdict = fund_err(M, v[u].x, v[u].y, v[u].z, v[u].w) <= thr2 ? 1 : -1;
                    }
                }
                cnt += verdict > 0;
            }
        }
        __syncthreads();
        const int nu = min(nund, kUndCap);
        for (int t = threadIdx.x; t < nu; t += 256) {
            const float4 w = pts[und[t]];
            int verdict = fund_filter(M, g64, w.x, w.y, w.z, w.w);
            if (verdict == 0) verdict = fund_err(M, w.x, w.y, w.z, w.w) <= thr2 ? 1 : -1;
            cnt += verdict > 0;
        }
        for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) counts[gs] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}
}  // namespace

// F scoring inside a RANSAC run whose earlier iterations have been replayed (result[4 b + 2] = best count so far of problem
// b): the matrix-core tier alone gives every model of list entries [nstart, nlist) an upper bound of its count; only the
// contenders -- bound above the best count -- are counted exactly.  counts[slot] ends as the exact count for contenders and
// as an upper bound <= the best count for everybody else, which OpenCV's loop (good > maxGoodCount) cannot tell apart.
// clist: batch x Hs ints, ncl: 17 batch + 1 ints (scratch: list lengths, one per 16 ints, then their prefix sums).
int vo_launch_score_fund_pruned(const double* models, const int32_t* list, const int32_t* nstart, const int32_t* nlist, int Hs,
                                const void* pts, const int32_t* n_dev, int N, int batch, float thr2, int32_t* counts,
                                int max_slots, const float* fund_bounds, const int32_t* result, int32_t* clist, int32_t* ncl,
                                hipStream_t s)
{
    if (Hs == 0 || N == 0 || batch == 0) return VO_OK;
    VO_CHECK_ARG(batch <= 65535, VO_ESIZE, "score: batch=%d too large", batch);
    VO_CHECK_ARG(fund_bounds != nullptr, VO_EINVAL, "score: the pruned F stage needs the per-problem coordinate bounds");
    if (max_slots <= 0 || max_slots > Hs) max_slots = Hs;
    ScoreArgs a;
    a.models = models; a.valid = nullptr; a.list = list; a.nstart = nstart; a.nlist = nlist; a.pts = pts; a.img = nullptr;
    a.n_dev = n_dev; a.counts = counts; a.H = Hs; a.N = N; a.thr2 = thr2; a.bounds = fund_bounds;
    a.chunk = kMfChunk;
    a.fx = a.fy = a.cx = a.cy = 0;
    const int nchunks = vo_ceil_div(N, kMfChunk), xblocks = vo_ceil_div(max_slots, kThreads);
    const int sblocks = vo_ceil_div(max_slots, kThreads * VO_SCREEN_MB);            // the screening kernel's workgroups hold VO_SCREEN_MB blocks
    const long mblocks = (long)sblocks * batch;
    long gy = (VO_SCORE_MF_TARGET + mblocks - 1) / mblocks;
    if (gy > nchunks) gy = nchunks;
    if (gy < 1) gy = 1;
    VO_HIP_TRY(hipMemsetAsync(ncl, 0, (size_t)batch * kNclStride * sizeof(int32_t), s));
    hipLaunchKernelGGL(screen_fund_mfma_kernel, dim3(sblocks, (unsigned)gy, batch), dim3(kThreads), 0, s, a, (int)gy);
    VO_LAUNCH_CHECK();
    hipLaunchKernelGGL(screen_finalize_kernel, dim3(xblocks < 64 ? xblocks : 64, batch), dim3(256), 0, s, list, nstart, nlist, Hs,
                       n_dev, N, result, counts, clist, ncl);
    VO_LAUNCH_CHECK();
    // the contenders, exactly (count_contenders_kernel above), as one index space over the batch
    hipLaunchKernelGGL(contender_prefix_kernel, dim3(1), dim3(1024), 0, s, ncl, batch, ncl + (size_t)batch * kNclStride);
    VO_LAUNCH_CHECK();
    hipLaunchKernelGGL(count_contenders_kernel, dim3(8192), dim3(256), 0, s, models, clist, ncl + (size_t)batch * kNclStride, batch, Hs, (const float4*)pts,
                       n_dev, N, thr2, fund_bounds, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_mask_best(int kind, const double* models, int slots_per_problem, int slots_per_sample,
                        const double* aux, int aux_w, double* best_aux, const int32_t* result, const void* pts,
                        const float* img, const int32_t* n_dev, int N, int batch, const double* K4, float thr2,
                        uint8_t* mask, double* best_model, hipStream_t s)
{
    if (batch == 0) return VO_OK;
    MaskArgs a;
    a.models = models; a.result = result; a.pts = pts; a.img = (const float2*)img; a.n_dev = n_dev; a.mask = mask;
    a.best_model = best_model; a.aux = aux; a.best_aux = best_aux; a.aux_w = aux_w;
    a.slots_per_problem = slots_per_problem; a.slots_per_sample = slots_per_sample; a.N = N; a.thr2 = thr2; a.direct = 0;
    a.fx = K4 ? K4[0] : 0; a.fy = K4 ? K4[1] : 0; a.cx = K4 ? K4[2] : 0; a.cy = K4 ? K4[3] : 0;
    const dim3 grid(vo_ceil_div(N > 0 ? N : 1, 256), batch);
    if (kind == kFund) hipLaunchKernelGGL(mask_best_kernel<kFund>, grid, dim3(256), 0, s, a);
    else if (kind == kEss) hipLaunchKernelGGL(mask_best_kernel<kEss>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mask_best_kernel<kPnp>, grid, dim3(256), 0, s, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_mask_direct(int kind, const double* best_model, const int32_t* result, const void* pts, const float* img,
                          const int32_t* n_dev, int N, int batch, const double* K4, float thr2, uint8_t* mask, hipStream_t s)
{
    if (batch == 0) return VO_OK;
    MaskArgs a;
    a.models = best_model; a.result = result; a.pts = pts; a.img = (const float2*)img; a.n_dev = n_dev; a.mask = mask;
    a.best_model = nullptr; a.aux = nullptr; a.best_aux = nullptr; a.aux_w = 0;
    a.slots_per_problem = 1; a.slots_per_sample = 1; a.N = N; a.thr2 = thr2; a.direct = 1;
    a.fx = K4 ? K4[0] : 0; a.fy = K4 ? K4[1] : 0; a.cx = K4 ? K4[2] : 0; a.cy = K4 ? K4[3] : 0;
    const dim3 grid(vo_ceil_div(N > 0 ? N : 1, 256), batch);
    if (kind == kFund) hipLaunchKernelGGL(mask_best_kernel<kFund>, grid, dim3(256), 0, s, a);
    else if (kind == kEss) hipLaunchKernelGGL(mask_best_kernel<kEss>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mask_best_kernel<kPnp>, grid, dim3(256), 0, s, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_select_owned(int kind, const double* models, int slots_per_problem, int slots_per_sample, const double* aux,
                           int aux_w, double* best_aux, const int32_t* result, int batch, int own_lo, int own_hi,
                           double* best_model, hipStream_t s)
{
    if (batch == 0) return VO_OK;
    const int mw = kind == kPnp ? 12 : 9;
    hipLaunchKernelGGL(select_owned_kernel, dim3(vo_ceil_div(batch, 64)), dim3(64), 0, s, models, aux, result, slots_per_problem,
                       slots_per_sample, mw, aux_w, own_lo, own_hi, batch, best_model, best_aux);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

#define SCORE_ARG_CHECKS(name)                                                                    \
    VO_CHECK_ARG(H >= 0 && N >= 0, VO_EINVAL, name ": negative size");                            \
    if (H == 0) return VO_OK;                                                                     \
    VO_CHECK_ARG(models && counts, VO_EINVAL, name ": null models/counts pointer");               \
    VO_CHECK_ARG(vo_ceil_div(H, kThreads) <= 2147483647 / kThreads, VO_ESIZE, name ": H too large")

extern "C" int vo_score_fund(const double* models, const uint8_t* valid, int H, const float* pts, int N,
                             float thr2, int32_t* counts, void* stream)
{
    SCORE_ARG_CHECKS("vo_score_fund");
    hipStream_t s = (hipStream_t)stream;
    int rc = vo_launch_init_counts(valid, nullptr, 1, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(pts && ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_score_fund: pts null or not 16-byte aligned");
    return vo_launch_score(kFund, models, valid, nullptr, nullptr, nullptr, H, pts, nullptr, nullptr, N, 1, nullptr, thr2, counts, H, s);
}

namespace {
__global__ void outliers_to_bound_kernel(int32_t* counts, int H, int N)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H) counts[i] = N - counts[i];
}
}  // namespace

extern "C" int vo_score_fund_bound(const double* models, int H, const float* pts, int N, float thr2, int32_t* bound, void* stream)
{
    VO_CHECK_ARG(H >= 0 && N >= 0, VO_EINVAL, "vo_score_fund_bound: negative size");
    if (H == 0) return VO_OK;
    VO_CHECK_ARG(models && bound, VO_EINVAL, "vo_score_fund_bound: null pointer");
    hipStream_t s = (hipStream_t)stream;
    VO_HIP_TRY(hipMemsetAsync(bound, 0, (size_t)H * sizeof(int32_t), s));
    if (N == 0) return VO_OK;
    VO_CHECK_ARG(pts && ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_score_fund_bound: pts null or not 16-byte aligned");
    ScoreArgs a;
    a.models = models; a.valid = nullptr; a.list = nullptr; a.nstart = nullptr; a.nlist = nullptr; a.pts = pts; a.img = nullptr;
    a.n_dev = nullptr; a.counts = bound; a.H = H; a.N = N; a.thr2 = thr2; a.bounds = nullptr;
    a.chunk = kMfChunk;
    a.fx = a.fy = a.cx = a.cy = 0;
    const int nchunks = vo_ceil_div(N, kMfChunk), xblocks = vo_ceil_div(H, kThreads * VO_SCREEN_MB);
    long gy = (VO_SCORE_MF_TARGET + xblocks - 1) / xblocks;
    if (gy > nchunks) gy = nchunks;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(screen_fund_mfma_kernel, dim3(xblocks, (unsigned)gy, 1), dim3(kThreads), 0, s, a, (int)gy);
    VO_LAUNCH_CHECK();
    hipLaunchKernelGGL(outliers_to_bound_kernel, dim3(vo_ceil_div(H, 256)), dim3(256), 0, s, bound, H, N);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_score_ess(const double* models, const uint8_t* valid, int H, const double* pts, int N,
                            float thr2, int32_t* counts, void* stream)
{
    SCORE_ARG_CHECKS("vo_score_ess");
    hipStream_t s = (hipStream_t)stream;
    int rc = vo_launch_init_counts(valid, nullptr, 1, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(pts != nullptr, VO_EINVAL, "vo_score_ess: null pts");
    return vo_launch_score(kEss, models, valid, nullptr, nullptr, nullptr, H, pts, nullptr, nullptr, N, 1, nullptr, thr2, counts, H, s);
}

extern "C" int vo_score_pnp(const double* models, const uint8_t* valid, int H, const float* obj,
                            const float* img, const double* K4_host, int N, float thr2, int32_t* counts,
                            void* stream)
{
    SCORE_ARG_CHECKS("vo_score_pnp");
    hipStream_t s = (hipStream_t)stream;
    int rc = vo_launch_init_counts(valid, nullptr, 1, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(obj && img && K4_host, VO_EINVAL, "vo_score_pnp: null pointer");
    VO_CHECK_ARG(((uintptr_t)img & 7) == 0, VO_EINVAL, "vo_score_pnp: img must be 8-byte aligned");
    return vo_launch_score(kPnp, models, valid, nullptr, nullptr, nullptr, H, obj, img, nullptr, N, 1, K4_host, thr2, counts, H, s);
}

extern "C" int vo_mask_fund(const double* model, const float* pts, int N, float thr2, uint8_t* mask, void* stream)
{
    VO_CHECK_ARG(N >= 0, VO_EINVAL, "vo_mask_fund: negative size");
    if (N == 0) return VO_OK;
    VO_CHECK_ARG(model && pts && mask && ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_mask_fund: null/unaligned pointer");
    hipLaunchKernelGGL(mask_fund_kernel, dim3(vo_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, model,
                       (const float4*)pts, N, thr2, mask);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_mask_ess(const double* model, const double* pts, int N, float thr2, uint8_t* mask, void* stream)
{
    VO_CHECK_ARG(N >= 0, VO_EINVAL, "vo_mask_ess: negative size");
    if (N == 0) return VO_OK;
    VO_CHECK_ARG(model && pts && mask, VO_EINVAL, "vo_mask_ess: null pointer");
    hipLaunchKernelGGL(mask_ess_kernel, dim3(vo_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, model, pts,
                       N, thr2, mask);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_mask_pnp(const double* model, const float* obj, const float* img, const double* K4_host, int N,
                           float thr2, uint8_t* mask, void* stream)
{
    VO_CHECK_ARG(N >= 0, VO_EINVAL, "vo_mask_pnp: negative size");
    if (N == 0) return VO_OK;
    VO_CHECK_ARG(model && obj && img && mask && K4_host && ((uintptr_t)img & 7) == 0, VO_EINVAL,
                 "vo_mask_pnp: null/unaligned pointer");
    hipLaunchKernelGGL(mask_pnp_kernel, dim3(vo_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, model, obj,
                       (const float2*)img, N, K4_host[0], K4_host[1], K4_host[2], K4_host[3], thr2, mask);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_lmeds_median_fund(const double* models, const uint8_t* valid, int H, const float* pts, int N,
                                    float* medians, void* stream)
{
    VO_CHECK_ARG(H >= 0 && N >= 1 && N <= kLmedsMaxPoints, VO_EINVAL,
                 "vo_lmeds_median_fund: N = %d outside 1..%d (LMedS serves the 8..14 point case)", N, kLmedsMaxPoints);
    if (H == 0) return VO_OK;
    VO_CHECK_ARG(models && pts && medians && ((uintptr_t)pts & 15) == 0, VO_EINVAL,
                 "vo_lmeds_median_fund: null/unaligned pointer");
    hipLaunchKernelGGL(lmeds_median_fund_kernel, dim3(vo_ceil_div(H, 64)), dim3(64), 0, (hipStream_t)stream, models,
                       valid, H, (const float4*)pts, N, medians);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_small_prepare(const int32_t* n_match, int batch, int H, int32_t* n_lmeds, int32_t* n_small, int32_t* samples,
                            hipStream_t s)
{
    if (batch == 0) return VO_OK;
    if (samples)
        hipLaunchKernelGGL(small_identity_kernel, dim3(vo_ceil_div(batch, 64)), dim3(64), 0, s, n_match, batch, H, samples);
    else
        hipLaunchKernelGGL(small_prepare_kernel, dim3(vo_ceil_div(batch, 64)), dim3(64), 0, s, n_match, batch, H, n_lmeds, n_small,
                           samples);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_small_finish(const double* models, const int32_t* nmodels, int H, const float* pts4, int N,
                           const int32_t* n_small, int batch, int32_t* result, double* F, uint8_t* mask, double* models7,
                           int32_t* nmodels7, hipStream_t s)
{
    if (batch == 0) return VO_OK;
    hipLaunchKernelGGL(small_finish_kernel, dim3(batch), dim3(64), 0, s, models, nmodels, H, (const float4*)pts4, N, n_small,
                       result, F, mask, models7, nmodels7);
    VO_LAUNCH_CHECK();
    return VO_OK;
}
