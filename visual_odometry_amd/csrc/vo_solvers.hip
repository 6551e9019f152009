// vo_solvers.hip -- batched minimal solvers, one RANSAC sample per lane, f64 throughout.
// They stand in for the runKernel callbacks behind cv2.findFundamentalMat (run7Point),
// cv2.findEssentialMat (Nister five-point) and cv.solvePnPRansac(SOLVEPNP_P3P) (Gao P3P + ranking)
// -- reference call sites src/utils/geometry.py:39,44 and src/pnp.py:16-26; algorithms as
// laid out in SURVEY.md Appendix A.3-A.6 (published methods: Hartley & Zisserman 7-point,
// Nister PAMI 2004, Gao et al. PAMI 2003, Horn's quaternion absolute orientation, one-sided
// Jacobi SVD, Durand-Kerner).
//
// Reproducibility: only IEEE + - * / sqrt and the vodm:: elementary functions are used, in a
// fixed operation order and without FMA contraction, so a scalar CPU evaluation of the same
// formulas gives the same bits.  Per-lane matrices live in LDS, element-major ([element][lane]),
// so dynamic indexing costs no scratch and no bank conflicts.
#include "vo_common.h"
#include "vo_detmath.h"
#include "vo_fivept_gen.h"
#include <float.h>

namespace {

constexpr int kLanes = 64;          // one wave per workgroup, one sample per lane
constexpr double kPi = 3.1415926535897932384626433832795;

// Compacted list of the model slots that hold a model, per problem: the scoring kernel then
// spends no lanes on the empty slots of samples with fewer than `width` solutions.  The order of
// the list depends on atomic arrival order; nothing downstream depends on it (counts land in
// their slot).  Called by every lane of the wave (n = 0 for lanes without a sample).  The
// per-problem counters of a batch share one or two cache lines, i.e. one or two L2 channels, so one
// atomic per lane serialises the whole launch there: a wave whose lanes all belong to one problem
// (the normal case) reserves its range with a single atomic instead.
__device__ __forceinline__ void append_slots(int32_t* list, int32_t* nlist, int b, int local_h, int per,
                                             int width, int n)
{
    if (!list) return;                                   // wave-uniform
    const int lane = threadIdx.x & 63;
    const int b0 = __shfl(b, __ffsll((long long)__ballot(n > 0)) - 1);
    int base;
    if (__all(n <= 0 || b == b0)) {
        int incl = n;                                    // inclusive scan of n over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int total = __shfl(incl, 63);
        if (total == 0) return;
        int start = 0;
        if (lane == 0) start = atomicAdd(nlist + b0, total);
        base = __shfl(start, 0) + incl - n;
    } else {
        base = n > 0 ? atomicAdd(nlist + b, n) : 0;
    }
    for (int k = 0; k < n; ++k) list[(size_t)b * per * width + base + k] = local_h * width + k;
}

struct SV {                         // element-major per-lane view into LDS
    double* p;
    __device__ __forceinline__ double& operator[](int i) const { return p[i * kLanes]; }
    __device__ __forceinline__ SV at(int off) const { return SV{p + off * kLanes}; }
};

// One-sided Jacobi SVD of the m x n matrix whose columns are the n rows of At (length m);
// rows n..n1-1 are completed to an orthonormal basis from a fixed +-1/m pattern by
// Gram-Schmidt.  On exit rows of At = left singular vectors (sorted, descending), Vt = V^T.
template <bool WANT_V>
__device__ void jacobi_svd(SV At, int m, int n, int n1, SV W, SV Vt)
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    const int max_iter = m > 30 ? m : 30;
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { const double t = At[i * m + k]; sd += t * t; }
        W[i] = sd;
        if (WANT_V) {
            for (int k = 0; k < n; ++k) Vt[i * n + k] = 0;
            Vt[i * n + i] = 1;
        }
    }
    for (int iter = 0; iter < max_iter; ++iter) {
        bool changed = false;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                const SV Ai = At.at(i * m), Aj = At.at(j * m);
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; ++k) {
                    const double t0 = c * Ai[k] + s * Aj[k];
                    const double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                if (WANT_V) {
                    const SV Vi = Vt.at(i * n), Vj = Vt.at(j * n);
                    for (int k = 0; k < n; ++k) {
                        const double t0 = c * Vi[k] + s * Vj[k];
                        const double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { const double t = At[i * m + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; ++i) {
        int j = i;
        for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (int k = 0; k < m; ++k) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
            if (WANT_V)
                for (int k = 0; k < n; ++k) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
        }
    }
    unsigned long long rs = 0x12345678ULL;
    for (int i = 0; i < n1; ++i) {
        double sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
            const double val0 = 1. / m;
            for (int k = 0; k < m; ++k) {
                rs = (unsigned long long)(unsigned)rs * 4164903690u + (unsigned)(rs >> 32);
                At[i * m + k] = ((unsigned)rs & 256) != 0 ? val0 : -val0;
            }
            for (int it = 0; it < 2; ++it)
                for (int j = 0; j < i; ++j) {
                    sd = 0;
                    for (int k = 0; k < m; ++k) sd += At[i * m + k] * At[j * m + k];
                    double asum = 0;
                    for (int k = 0; k < m; ++k) {
                        const double t = At[i * m + k] - sd * At[j * m + k];
                        At[i * m + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; ++k) At[i * m + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < m; ++k) { const double t = At[i * m + k]; sd += t * t; }
            sd = sqrt(sd);
        }
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; ++k) At[i * m + k] *= s;
    }
}

// The same algorithm (no V) on a matrix held in registers: every loop over rows and elements is
// unrolled so that all indices are compile-time constants; data-dependent control flow (rotation
// skipped, convergence, degenerate rows) is per-lane predication.  Operation order is that of
// jacobi_svd above, element for element, so both produce the same bits.  A lane that has converged
// rides along with the sweeps other lanes still need: with nothing left to rotate they change nothing.
template <int M, int N, int N1>
__device__ __forceinline__ void jacobi_rows_reg(double (&A)[N1][M], double (&W)[N1])
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    constexpr int max_iter = M > 30 ? M : 30;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += A[i][k] * A[i][k];
        W[i] = sd;
    }
    // Pair order is OpenCV's: (0,1) (0,2) .. (0,N-1) (1,2) ..  To keep the sweep small enough for the
    // instruction cache only the pairs (0,j) exist as code; after row i has met all its partners the
    // rows are rotated up by one (register moves), so the next pivot row sits in slot 0 again and the
    // rows already done wrap to the end, where the shrinking partner count leaves them alone.  N
    // rotations restore the order.
#pragma unroll 1
    for (int iter = 0; iter < max_iter; ++iter) {
        bool changed = false;
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 1; j < N; ++j) {
                if (j > N - 1 - i) continue;               // wave-uniform
                double a = W[0], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; ++k) p += A[0][k] * A[j][k];
                // OpenCV's test is |p| <= eps sqrt(a b).  The square root only matters within rounding distance of equality:
                // p^2 against eps^2 a b with a 2^-30 guard band decides everything else (both sides normal numbers, so neither
                // product has lost bits to underflow; NaN and Inf fail both comparisons and take the exact test)
                const double pp = p * p, rhs = (eps * eps) * (a * b);
                bool rotate;
                if (pp > rhs * (1. + 0x1p-30) && rhs >= minval) rotate = true;
                else if (pp < rhs * (1. - 0x1p-30) && pp >= minval) rotate = false;
                else rotate = !(fabs(p) <= eps * sqrt(a * b));
                if (rotate) {
                    p *= 2;
                    const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                    double c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = sqrt(delta / gamma);
                        c = p / (gamma * s * 2);
                    } else {
                        c = sqrt((gamma + beta) / (gamma * 2));
                        s = p / (gamma * c * 2);
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const double t0 = c * A[0][k] + s * A[j][k];
                        const double t1 = -s * A[0][k] + c * A[j][k];
                        A[0][k] = t0; A[j][k] = t1;
                        a += t0 * t0; b += t1 * t1;
                    }
                    W[0] = a; W[j] = b;
                    changed = true;
                }
            }
            const double w0 = W[0];
#pragma unroll
            for (int r = 0; r < N - 1; ++r) W[r] = W[r + 1];
            W[N - 1] = w0;
#pragma unroll
            for (int k = 0; k < M; ++k) {
                const double t = A[0][k];
#pragma unroll
                for (int r = 0; r < N - 1; ++r) A[r][k] = A[r + 1][k];
                A[N - 1][k] = t;
            }
        }
        if (!__any(changed)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += A[i][k] * A[i][k];
        W[i] = sqrt(sd);
    }
    // selection sort, descending: row i <-> the first largest of rows i..N-1
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < N; ++k)
            if (j == k) {
                double t = W[i]; W[i] = W[k]; W[k] = t;
#pragma unroll
                for (int e = 0; e < M; ++e) { t = A[i][e]; A[i][e] = A[k][e]; A[k][e] = t; }
            }
    }
    unsigned long long rs = 0x12345678ULL;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
        double sd = i < N ? W[i] : 0;
#pragma unroll 1
        for (int ii = 0; ii < 100 && __any(sd <= minval); ++ii) {
            // the +-1/M pattern is the same for every lane; lanes whose row is fine keep theirs, but
            // must draw exactly as many numbers as a scalar run would: none.  A lane that is not
            // degenerate here never consumes from rs, so rs has to be per lane.
            if (sd <= minval) {
                const double val0 = 1. / M;
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    rs = (unsigned long long)(unsigned)rs * 4164903690u + (unsigned)(rs >> 32);
                    A[i][k] = ((unsigned)rs & 256) != 0 ? val0 : -val0;
                }
#pragma unroll 1
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int j = 0; j < i; ++j) {
                        double d = 0;
#pragma unroll
                        for (int k = 0; k < M; ++k) d += A[i][k] * A[j][k];
                        double asum = 0;
#pragma unroll
                        for (int k = 0; k < M; ++k) {
                            const double t = A[i][k] - d * A[j][k];
                            A[i][k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                        for (int k = 0; k < M; ++k) A[i][k] *= asum;
                    }
                double q = 0;
#pragma unroll
                for (int k = 0; k < M; ++k) q += A[i][k] * A[i][k];
                sd = sqrt(q);
            }
        }
        const double sc = sd > minval ? 1 / sd : 0.;
#pragma unroll
        for (int k = 0; k < M; ++k) A[i][k] *= sc;
    }
}

// a0 x^3 + a1 x^2 + a2 x + a3 = 0 (Cardano / trigonometric form).  returns #real roots
__device__ int solve_cubic(double a0, double a1, double a2, double a3, double* r)
{
    double x0 = 0, x1 = 0, x2 = 0;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = vodm::acos_(R / sqrt(Qcubed));
            const double sqrtQ = sqrt(Q);
            const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * vodm::cos_(t1) - t2;
            x1 = t0 * vodm::cos_(t1 + (2. * kPi / 3)) - t2;
            x2 = t0 * vodm::cos_(t1 + (4. * kPi / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * vodm::cbrt_(R) - a1 / 3; x1 = vodm::cbrt_(R) - a1 / 3; }
            else { x0 = 2 * vodm::cbrt_(-R) - a1 / 3; x1 = -vodm::cbrt_(-R) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            d = sqrt(-d);
            double e = vodm::cbrt_(d + fabs(R));
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    r[0] = x0; r[1] = x1; r[2] = x2;
    return n;
}

// ------------------------------------------------------------------------------ 7-point
__global__ __launch_bounds__(kLanes)
void solve_7pt_kernel(const int32_t* __restrict__ samples, int H, const float4* __restrict__ pts, int N,
                      double* __restrict__ models, int32_t* __restrict__ nmodels,
                      int per, int32_t* __restrict__ list, int32_t* __restrict__ nlist, const int32_t* __restrict__ n_dev,
                      int h0, int hlen, const int32_t* __restrict__ prev)
{
    // thread -> (problem b, iteration h0 + i) for the iteration block [h0, h0 + hlen) of every problem;
    // a problem whose sequential loop already ended (prev = its replay result, niters <= h0) is skipped
    const int tix = blockIdx.x * kLanes + threadIdx.x;
    bool live = tix < (H / per) * hlen;
    const int b = live ? tix / hlen : 0;
    const int h = b * per + h0 + (tix - b * hlen);
    if (prev && prev[4 * b + 3] <= h0) live = false;
    if (live && samples[(size_t)h * 7] < 0) live = false;     // iteration without a sample (getSubset gave up)
    if (!__any(live)) return;                 // problem this sample belongs to
    const int nb_pts = n_dev ? min(n_dev[b], N) : N;
    pts += (size_t)b * N;
    // the 7 x 9 epipolar constraint rows and, after the SVD, the two null-space rows: all in VGPRs
    double A[9][9], W[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        W[i] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) A[i][k] = 0;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int idx = live ? samples[(size_t)h * 7 + i] : 0;
        idx = max(min(idx, nb_pts - 1), 0);
        const float4 p = pts[idx];
        const double x0 = p.x, y0 = p.y, x1 = p.z, y1 = p.w;
        A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
        A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
        A[i][6] = x0; A[i][7] = y0; A[i][8] = 1;
    }
    jacobi_rows_reg<9, 7, 9>(A, W);

    double f1[9], f2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { f2[i] = A[8][i]; f1[i] = A[7][i] - f2[i]; }
    double c[4], r[3];
    double t0, t1, t2;
    t0 = f2[4] * f2[8] - f2[5] * f2[7];
    t1 = f2[3] * f2[8] - f2[5] * f2[6];
    t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) -
           f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) -
           f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    int n = solve_cubic(c[0], c[1], c[2], c[3], r);
    if (n < 1 || n > 3) n = 0;
    append_slots(list, nlist, b, h - b * per, per, 3, live ? n : 0);
    if (!live) return;
    nmodels[h] = n;
    for (int k = 0; k < n; ++k) {
        double* F = models + ((size_t)h * 3 + k) * 9;
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        double f8;
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; f8 = 1.; }
        else f8 = 0.;
#pragma unroll
        for (int i = 0; i < 8; ++i) F[i] = f1[i] * lambda + f2[i] * mu;
        F[8] = f8;
    }
}

// ------------------------------------------------------------------------------ quartic (MathWorld)
__device__ int solve_deg2(double a, double b, double c, double& x1, double& x2)
{
    const double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    const double inv_2a = 0.5 / a;
    if (delta == 0) { x1 = -b * inv_2a; x2 = x1; return 1; }
    const double sq = sqrt(delta);
    x1 = (-b + sq) * inv_2a;
    x2 = (-b - sq) * inv_2a;
    return 2;
}

__device__ int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            x0 = -d / c;
            return 1;
        }
        x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    const double inv_a = 1. / a;
    const double b_a = inv_a * b, b_a2 = b_a * b_a;
    const double c_a = inv_a * c, d_a = inv_a * d;
    const double Q = (3 * c_a - b_a2) / 9;
    const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { x0 = x1 = x2 = -b_a_3; return 3; }
        x0 = (2 * R > 0 ? vodm::cbrt_(2 * R) : __builtin_nan("")) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        const double theta = vodm::acos_(R / sqrt(-Q3));
        const double sqrt_Q = sqrt(-Q);
        x0 = 2 * sqrt_Q * vodm::cos_(theta / 3.0) - b_a_3;
        x1 = 2 * sqrt_Q * vodm::cos_((theta + 2 * kPi) / 3.0) - b_a_3;
        x2 = 2 * sqrt_Q * vodm::cos_((theta + 4 * kPi) / 3.0) - b_a_3;
        return 3;
    }
    const double AD = vodm::cbrt_(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}

__device__ int solve_deg4(double a, double b, double c, double d, double e, double* x)
{
    x[0] = x[1] = x[2] = x[3] = 0;
    if (a == 0) return solve_deg3(b, c, d, e, x[0], x[1], x[2]);
    const double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0 = 0, r1 = 0, r2 = 0;
    const int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    const double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    const double R = sqrt(R2);
    const double inv_R = 1. / R;
    int nb = 0;
    double D2, E2;
    if (R < 10E-12) {
        const double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            const double st = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * st;
            E2 = D2 - 4 * st;
        }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        const double D = sqrt(D2);
        nb = 2;
        const double D_2 = 0.5 * D;
        x[0] = R_2 + D_2 - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        const double E = sqrt(E2), E_2 = 0.5 * E;
        if (nb == 0) { x[0] = -R_2 + E_2 - b_4; x[1] = x[0] - E; nb = 2; }
        else { x[2] = -R_2 + E_2 - b_4; x[3] = x[2] - E; nb = 4; }
    }
    return nb;
}

// jacobi_svd<true>(At, 3, 3, 3, W, Vt) in registers: every index a compile-time constant, same pair order and operation
// order, so the bits agree.  On exit rows of At = left singular vectors (if COMPLETE: normalised, degenerate ones replaced
// exactly as the generic routine does), W descending, Vt = V^T.  A lane that has converged rides along with the sweeps
// other lanes still need: with nothing left to rotate it changes nothing.
template <bool COMPLETE>
__device__ __forceinline__ void jacobi3_svd(double (&At)[3][3], double (&W)[3], double (&Vt)[3][3])
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) sd += At[i][k] * At[i][k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < 3; ++k) Vt[i][k] = i == k ? 1. : 0.;
    }
#pragma unroll 1
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < 3; ++k) p += At[i][k] * At[j][k];
                if (!(fabs(p) <= eps * sqrt(a * b))) {
                    p *= 2;
                    const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                    double c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = sqrt(delta / gamma);
                        c = p / (gamma * s * 2);
                    } else {
                        c = sqrt((gamma + beta) / (gamma * 2));
                        s = p / (gamma * c * 2);
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double t0 = c * At[i][k] + s * At[j][k];
                        const double t1 = -s * At[i][k] + c * At[j][k];
                        At[i][k] = t0; At[j][k] = t1;
                        a += t0 * t0; b += t1 * t1;
                    }
                    W[i] = a; W[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double t0 = c * Vt[i][k] + s * Vt[j][k];
                        const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                        Vt[i][k] = t0; Vt[j][k] = t1;
                    }
                }
            }
        if (!__any(changed)) break;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < 3; ++k)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < 3; ++k)
            if (j == k) {
                double t = W[i]; W[i] = W[k]; W[k] = t;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    t = At[i][e]; At[i][e] = At[k][e]; At[k][e] = t;
                    t = Vt[i][e]; Vt[i][e] = Vt[k][e]; Vt[k][e] = t;
                }
            }
    }
    if (!COMPLETE) return;
    unsigned long long rs = 0x12345678ULL;              // per lane: only degenerate rows draw from it
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double sd = W[i];
#pragma unroll 1
        for (int ii = 0; ii < 100 && __any(sd <= minval); ++ii) {
            if (sd <= minval) {
                const double val0 = 1. / 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    rs = (unsigned long long)(unsigned)rs * 4164903690u + (unsigned)(rs >> 32);
                    At[i][k] = ((unsigned)rs & 256) != 0 ? val0 : -val0;
                }
#pragma unroll 1
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int j = 0; j < i; ++j) {
                        double d = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) d += At[i][k] * At[j][k];
                        double asum = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const double t = At[i][k] - d * At[j][k];
                            At[i][k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) At[i][k] *= asum;
                    }
                double q = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) q += At[i][k] * At[i][k];
                sd = sqrt(q);
            }
        }
        const double sc = sd > minval ? 1 / sd : 0.;
#pragma unroll
        for (int k = 0; k < 3; ++k) At[i][k] *= sc;
    }
}

// ------------------------------------------------------------------------------ Rodrigues
__device__ void rodrigues_v2m(const double* r_in, double* R)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double s, c;
    vodm::sincos(theta, &s, &c);
    const double c1 = 1. - c;
    const double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = (c * I[k] + c1 * rrt[k]) + s * r_x[k];
}

__device__ void rodrigues_m2v(const double* Rin, double* rvec)
{
    double At[3][3], W[3], Vt[3][3];
    bool in_range = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) in_range &= (Rin[k] > -100 && Rin[k] < 100);
    // lanes whose matrix is out of range ride through the (wave-uniform) SVD loops on a harmless identity
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) At[i][j] = in_range ? Rin[j * 3 + i] : (i == j ? 1. : 0.);
    jacobi3_svd<true>(At, W, Vt);
    if (!in_range) { rvec[0] = rvec[1] = rvec[2] = 0; return; }
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += At[k][i] * Vt[k][j];
            R[i * 3 + j] = s;
        }
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    const double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = vodm::acos_(c);
    if (s < 1e-5) {
        if (c > 0) { x = y = z = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; x = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; y = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    rvec[0] = x; rvec[1] = y; rvec[2] = z;
}

__device__ __forceinline__ void project_one(const double* R, const double* t, double fx, double fy,
                                            double cx, double cy, double X, double Y, double Z,
                                            double* u, double* v)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z != 0. ? 1. / z : 1;
    x *= z; y *= z;
    *u = x * fx + cx;
    *v = y * fy + cy;
}

// ------------------------------------------------------------------------------ P3P
// cyclic Jacobi eigen-decomposition of a symmetric 4x4 (Numerical-Recipes scheme)
__device__ void jacobi_4x4(double* A, double* D, double* U)
{
    double B[4], Z[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1. : 0.;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
#pragma unroll
    for (int i = 0; i < 4; ++i) D[i] = B[i];
#pragma unroll 1
    for (int iter = 0; iter < 50; ++iter) {
        const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return;
        const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                const double Aij = A[4 * i + j];
                const double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
                    A[4 * i + j] = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        const double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    A[4 * i + j] = 0.0;
                    const double c = 1.0 / sqrt(1 + t * t);
                    const double s = t * c;
                    const double tau = s / (1.0 + c);
#pragma unroll
                    for (int k = 0; k <= i - 1; ++k) {
                        const double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
#pragma unroll
                    for (int k = i + 1; k <= j - 1; ++k) {
                        const double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
#pragma unroll
                    for (int k = j + 1; k < 4; ++k) {
                        const double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { B[i] += Z[i]; D[i] = B[i]; Z[i] = 0; }
    }
}

// Horn's quaternion absolute orientation of 3 points: M_end = R * X + T
__device__ void p3p_align(const double M[3][3], const double* X, double* R, double* T)
{
    const double X0 = X[0], Y0 = X[1], Z0 = X[2], X1 = X[3], Y1 = X[4], Z1 = X[5], X2 = X[6], Y2 = X[7], Z2 = X[8];
    double Cs[3], Ce[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ce[i] = (M[0][i] + M[1][i] + M[2][i]) / 3;
    Cs[0] = (X0 + X1 + X2) / 3;
    Cs[1] = (Y0 + Y1 + Y2) / 3;
    Cs[2] = (Z0 + Z1 + Z2) / 3;
    double s[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[0 * 3 + j] = (X0 * M[0][j] + X1 * M[1][j] + X2 * M[2][j]) / 3 - Ce[j] * Cs[0];
        s[1 * 3 + j] = (Y0 * M[0][j] + Y1 * M[1][j] + Y2 * M[2][j]) / 3 - Ce[j] * Cs[1];
        s[2 * 3 + j] = (Z0 * M[0][j] + Z1 * M[1][j] + Z2 * M[2][j]) / 3 - Ce[j] * Cs[2];
    }
    double Qs[16], evs[4], U[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Qs[i] = 0;
    Qs[0 * 4 + 0] = s[0] + s[4] + s[8];
    Qs[1 * 4 + 1] = s[0] - s[4] - s[8];
    Qs[2 * 4 + 2] = s[4] - s[8] - s[0];
    Qs[3 * 4 + 3] = s[8] - s[0] - s[4];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    double q[4] = {U[0], U[4], U[8], U[12]};
    double ev_max = evs[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (evs[i] > ev_max) { ev_max = evs[i]; q[0] = U[0 + i]; q[1] = U[4 + i]; q[2] = U[8 + i]; q[3] = U[12 + i]; }
    const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    const double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    const double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0] = q02 + q12 - q22 - q32; R[1] = 2. * (q1_2 - q0_3); R[2] = 2. * (q1_3 + q0_2);
    R[3] = 2. * (q1_2 + q0_3); R[4] = q02 + q22 - q12 - q32; R[5] = 2. * (q2_3 - q0_1);
    R[6] = 2. * (q1_3 - q0_2); R[7] = 2. * (q2_3 + q0_1); R[8] = q02 + q32 - q12 - q22;
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = Ce[i] - (R[3 * i] * Cs[0] + R[3 * i + 1] * Cs[1] + R[3 * i + 2] * Cs[2]);
}

// distances |BC|,|AC|,|AB| ; cosines of BPC, APC, APB -> up to 4 (|PA|,|PB|,|PC|) triples
__device__ int p3p_lengths(double lengths[4][3], const double* distances, const double* cosines)
{
    const double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    const double inv_d22 = 1. / (distances[2] * distances[2]);
    const double a = inv_d22 * (distances[0] * distances[0]);
    const double b = inv_d22 * (distances[1] * distances[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    const double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    const double ab = a * b, a_2 = 2 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    const double a_4 = 4 * a;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    const double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    const int n = solve_deg4(A, B, C, D, E, roots);
    if (n == 0) return 0;
    int nb = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    const double inv_b0 = 1. / b0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= n) break;
        const double x = roots[i];
        if (x <= 0) continue;
        const double x2 = x * x;
        const double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) +
               pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
              r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) +
             pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        const double y = inv_b0 * b1;
        const double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        const double Z = distances[2] / sqrt(v);
        lengths[nb][0] = x * Z; lengths[nb][1] = y * Z; lengths[nb][2] = Z;
        nb++;
    }
    return nb;
}

__global__ __launch_bounds__(kLanes)
void solve_p3p_kernel(const int32_t* __restrict__ samples, int H, const float* __restrict__ obj,
                      const float2* __restrict__ img, int N, double fx, double fy, double cx, double cy,
                      double* __restrict__ models, double* __restrict__ rvecs, int32_t* __restrict__ nmodels,
                      int per, int32_t* __restrict__ list, int32_t* __restrict__ nlist, const int32_t* __restrict__ n_dev,
                      int h0, int hlen, const int32_t* __restrict__ prev)
{
    // thread -> (problem b, iteration h0 + i) for the iteration block [h0, h0 + hlen) of every problem;
    // a problem whose sequential loop already ended (prev = its replay result, niters <= h0) is skipped
    const int tix = blockIdx.x * kLanes + threadIdx.x;
    bool live = tix < (H / per) * hlen;
    const int b = live ? tix / hlen : 0;
    const int h = b * per + h0 + (tix - b * hlen);
    if (prev && prev[4 * b + 3] <= h0) live = false;
    if (live && samples[(size_t)h * 4] < 0) live = false;     // iteration without a sample (getSubset gave up)
    if (!__any(live)) return;
    const int nb_pts = n_dev ? min(n_dev[b], N) : N;
    obj += (size_t)b * N * 3;
    img += (size_t)b * N;

    const double ifx = 1. / fx, ify = 1. / fy;
    const double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double X[12], iu[4], iv[4], u[4], v[4], kk[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int idx = live ? samples[(size_t)h * 4 + i] : 0;
        idx = max(min(idx, nb_pts - 1), 0);
        const float2 uv = img[idx];
        iu[i] = uv.x; iv[i] = uv.y;
        // undistortPoints round trip: (u-cx)*(1/fx) stored as f32, then x*fx+cx
        const float nx = (float)((iu[i] - cx) * ifx), ny = (float)((iv[i] - cy) * ify);
        const double mu = nx * fx + cx, mv = ny * fy + cy;
        u[i] = inv_fx * mu - cx_fx;
        v[i] = inv_fy * mv - cy_fy;
        X[3 * i] = obj[(size_t)idx * 3]; X[3 * i + 1] = obj[(size_t)idx * 3 + 1]; X[3 * i + 2] = obj[(size_t)idx * 3 + 2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double norm = sqrt(u[i] * u[i] + v[i] * v[i] + 1);
        kk[i] = 1. / norm; u[i] *= kk[i]; v[i] *= kk[i];
    }
    const double X0 = X[0], Y0 = X[1], Z0 = X[2], X1 = X[3], Y1 = X[4], Z1 = X[5], X2 = X[6], Y2 = X[7], Z2 = X[8];
    const double X3 = X[9], Y3 = X[10], Z3 = X[11];
    double distances[3], cosines[3];
    distances[0] = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2) + (Z1 - Z2) * (Z1 - Z2));
    distances[1] = sqrt((X0 - X2) * (X0 - X2) + (Y0 - Y2) * (Y0 - Y2) + (Z0 - Z2) * (Z0 - Z2));
    distances[2] = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1) + (Z0 - Z1) * (Z0 - Z1));
    cosines[0] = u[1] * u[2] + v[1] * v[2] + kk[1] * kk[2];
    cosines[1] = u[0] * u[2] + v[0] * v[2] + kk[0] * kk[2];
    cosines[2] = u[0] * u[1] + v[0] * v[1] + kk[0] * kk[1];
    double lengths[4][3];
    const int n = p3p_lengths(lengths, distances, cosines);

    // rank exactly as the scalar definition does: stable insertion by the 4th-point error in
    // normalised coordinates, then stable insertion by the 4-point pixel reprojection error of
    // the Rodrigues round-tripped pose; only the winner is kept.
    double bestR[9], bestT[3], bestRv[3];
    double e3[4], e4[4];
    int order[4] = {0, 1, 2, 3};
    double Rs[4][9], Ts[4][3];
    int nb = 0;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        double M[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            M[j][0] = lengths[i][j] * u[j];
            M[j][1] = lengths[i][j] * v[j];
            M[j][2] = lengths[i][j] * kk[j];
        }
        double R[9], T[3];
        p3p_align(M, X, R, T);
        const double X3p = R[0] * X3 + R[1] * Y3 + R[2] * Z3 + T[0];
        const double Y3p = R[3] * X3 + R[4] * Y3 + R[5] * Z3 + T[1];
        const double Z3p = R[6] * X3 + R[7] * Y3 + R[8] * Z3 + T[2];
        const double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        e3[nb] = (mu3p - u[3]) * (mu3p - u[3]) + (mv3p - v[3]) * (mv3p - v[3]);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rs[nb][k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Ts[nb][k] = T[k];
        nb++;
    }
    // first ranking (indices only)
    for (int i = 1; i < nb; ++i)
        for (int j = i; j > 0 && e3[order[j - 1]] > e3[order[j]]; --j) {
            const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
        }
    // reprojection error of each candidate, in first-ranking order
    double rvs[4][3];
#pragma unroll 1
    for (int s = 0; s < nb; ++s) {
        const int o = order[s];
        double Rr[9];
        rodrigues_m2v(Rs[o], rvs[o]);
        rodrigues_v2m(rvs[o], Rr);
        double e = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double pu, pv;
            project_one(Rr, Ts[o], fx, fy, cx, cy, X[3 * i], X[3 * i + 1], X[3 * i + 2], &pu, &pv);
            const double du = iu[i] - pu, dv = iv[i] - pv;
            e += du * du;
            e += dv * dv;
        }
        e4[s] = e;
    }
    // second ranking: stable insertion over the positions of the first ranking
    int pos[4] = {0, 1, 2, 3};
    for (int i = 1; i < nb; ++i)
        for (int j = i; j > 0 && e4[pos[j - 1]] > e4[pos[j]]; --j) {
            const int tmp = pos[j]; pos[j] = pos[j - 1]; pos[j - 1] = tmp;
        }
    append_slots(list, nlist, b, h - b * per, per, 1, live && nb > 0 ? 1 : 0);
    if (!live) return;
    nmodels[h] = nb > 0 ? 1 : 0;
    if (nb > 0) {
        const int o = order[pos[0]];
#pragma unroll
        for (int k = 0; k < 3; ++k) { bestRv[k] = rvs[o][k]; bestT[k] = Ts[o][k]; }
        rodrigues_v2m(bestRv, bestR);
#pragma unroll
        for (int k = 0; k < 9; ++k) models[(size_t)h * 12 + k] = bestR[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { models[(size_t)h * 12 + 9 + k] = bestT[k]; rvecs[(size_t)h * 3 + k] = bestRv[k]; }
    }
}

// ------------------------------------------------------------------------------ five-point
// polynomials in (x,y,z) of total degree <= 3 over the 20 monomials, addressed through the
// exponent triple so that products accumulate in a fixed (lexicographic) order
struct Poly { double c[20]; };

__constant__ signed char kPolyIdx[4][4][4] = {
    // [i][j][k] -> slot (or -1 when i+j+k > 3); slots follow lexicographic (i,j,k)
    {{0, 1, 2, 3}, {4, 5, 6, -1}, {7, 8, -1, -1}, {9, -1, -1, -1}},
    {{10, 11, 12, -1}, {13, 14, -1, -1}, {15, -1, -1, -1}, {-1, -1, -1, -1}},
    {{16, 17, -1, -1}, {18, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}},
    {{19, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}}};

__device__ void p_zero(Poly& p) { for (int i = 0; i < 20; ++i) p.c[i] = 0; }

__device__ void p_mul(const Poly& a, const Poly& b, Poly& r)
{
    p_zero(r);
    for (int i = 0; i < 4; ++i) for (int j = 0; j + i < 4; ++j) for (int k = 0; k + i + j < 4; ++k) {
        const double av = a.c[kPolyIdx[i][j][k]];
        if (av == 0) continue;
        for (int l = 0; l + i < 4; ++l) for (int m = 0; m + j < 4; ++m) for (int n = 0; n + k < 4; ++n) {
            if (i + j + k + l + m + n > 3) continue;
            r.c[kPolyIdx[i + l][j + m][k + n]] += av * b.c[kPolyIdx[l][m][n]];
        }
    }
}

__device__ void p_axpy(Poly& r, double s, const Poly& a) { for (int i = 0; i < 20; ++i) r.c[i] += s * a.c[i]; }

// row order [x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1]
__constant__ signed char kMon[20][3] = {
    {3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
    {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

__device__ void p_row(const Poly& p, double* row)
{
    for (int m = 0; m < 20; ++m) row[m] = p.c[kPolyIdx[kMon[m][0]][kMon[m][1]][kMon[m][2]]];
}

__device__ void z_mul(const double* a, int da, const double* b, int db, double* r)
{
    for (int i = 0; i <= da + db; ++i) r[i] = 0;
    for (int i = 0; i <= da; ++i) for (int j = 0; j <= db; ++j) r[i + j] += a[i] * b[j];
}

// cv::solvePoly's default iteration cap (core.hpp: maxIters = 300), which is what five-point.cpp runs with.  The
// update almost never reaches an exact fixpoint, so in practice this many sweeps always run.
constexpr int kSolvePolyIters = 300;

// Durand-Kerner on ascending real coefficients; returns #roots
__device__ int solve_poly(const double* coeffs, int deg, double* re, double* im)
{
    int n = deg;
    for (; n > 1; --n)
        if (fabs(coeffs[n]) > DBL_EPSILON) break;
    double pr = 1, pi = 0;
    for (int i = 0; i < n; ++i) {
        re[i] = pr; im[i] = pi;
        const double nr = pr * 1 - pi * 1, ni = pr * 1 + pi * 1;
        pr = nr; pi = ni;
    }
    for (int iter = 0; iter < kSolvePolyIters; ++iter) {
        double maxdiff = 0;
        for (int i = 0; i < n; ++i) {
            const double xr = re[i], xi = im[i];
            double nr = coeffs[n], ni = 0, dr = coeffs[n], di = 0;
            for (int j = 0; j < n; ++j) {
                double tr = nr * xr - ni * xi + coeffs[n - j - 1];
                double ti = nr * xi + ni * xr;
                nr = tr; ni = ti;
                if (j != i) {
                    const double er = xr - re[j], ei = xi - im[j];
                    if (er != 0 || ei != 0) {
                        tr = dr * er - di * ei;
                        ti = dr * ei + di * er;
                        dr = tr; di = ti;
                    }
                }
            }
            const double t = 1. / (dr * dr + di * di);
            const double qr = (nr * dr + ni * di) * t, qi = (-nr * di + ni * dr) * t;
            re[i] = xr - qr; im[i] = xi - qi;
            const double ad = sqrt(qr * qr + qi * qi);
            if (ad > maxdiff) maxdiff = ad;
        }
        if (maxdiff <= 0) break;
    }
    for (int i = 0; i < n; ++i)
        if (fabs(im[i]) < 1e-100) im[i] = 0;
    return n;
}

// The same iteration for a polynomial whose degree is exactly N (leading coefficient above DBL_EPSILON), roots in
// registers, every index a compile-time constant; operation order as in solve_poly, so the bits agree.  A lane that
// has reached an exact fixpoint (or NaNs) rides along with the sweeps other lanes still need: its roots no longer change.
template <int N>
__device__ __forceinline__ void solve_poly_fixed(const double (&c)[N + 1], double (&re)[N], double (&im)[N])
{
    double pr = 1, pi = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        re[i] = pr; im[i] = pi;
        const double nr = pr * 1 - pi * 1, ni = pr * 1 + pi * 1;
        pr = nr; pi = ni;
    }
#pragma unroll 1
    for (int iter = 0; iter < kSolvePolyIters; ++iter) {
        double maxdiff = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double xr = re[i], xi = im[i];
            double nr = c[N], ni = 0, dr = c[N], di = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double tr = nr * xr - ni * xi + c[N - j - 1];
                double ti = nr * xi + ni * xr;
                nr = tr; ni = ti;
                if (j != i) {
                    const double er = xr - re[j], ei = xi - im[j];
                    tr = dr * er - di * ei;
                    ti = dr * ei + di * er;
                    const bool distinct = er != 0 || ei != 0;
                    dr = distinct ? tr : dr; di = distinct ? ti : di;
                }
            }
            const double t = 1. / (dr * dr + di * di);
            const double qr = (nr * dr + ni * di) * t, qi = (-nr * di + ni * dr) * t;
            re[i] = xr - qr; im[i] = xi - qi;
            const double ad = sqrt(qr * qr + qi * qi);
            if (ad > maxdiff) maxdiff = ad;
        }
        if (!__any(maxdiff > 0)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (fabs(im[i]) < 1e-100) im[i] = 0;
}

// Right singular vector of the smallest singular value of a 3 x 3 matrix (rows of At = columns of the matrix): the last
// row of V^T after the descending sort.  The completion of degenerate left singular vectors is skipped: it never touches V.
__device__ __forceinline__ void jacobi3_last_vt(double (&At)[3][3], double (&out)[3])
{
    double W[3], Vt[3][3];
    jacobi3_svd<false>(At, W, Vt);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = Vt[2][k];
}

__global__ __launch_bounds__(kLanes)
void solve_5pt_kernel(const int32_t* __restrict__ samples, int H, const double* __restrict__ pts, int N,
                      double* __restrict__ models, int32_t* __restrict__ nmodels,
                      int per, int32_t* __restrict__ list, int32_t* __restrict__ nlist, const int32_t* __restrict__ n_dev,
                      int h0, int hlen, const int32_t* __restrict__ prev)
{
    // thread -> (problem b, iteration h0 + i) for the iteration block [h0, h0 + hlen) of every problem;
    // a problem whose sequential loop already ended (prev = its replay result, niters <= h0) is skipped
    const int tix = blockIdx.x * kLanes + threadIdx.x;
    bool live = tix < (H / per) * hlen;
    const int b = live ? tix / hlen : 0;
    const int h = b * per + h0 + (tix - b * hlen);
    if (prev && prev[4 * b + 3] <= h0) live = false;
    if (live && samples[(size_t)h * 5] < 0) live = false;     // iteration without a sample (getSubset gave up)
    if (!__any(live)) return;
    const int nb_pts = n_dev ? min(n_dev[b], N) : N;
    pts += (size_t)b * N * 4;
    double Nb[4][9];
    {
        // null space of the 5 x 9 epipolar system: the rows the Jacobi SVD completes the basis with (registers)
        double A9[9][9], W9[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int k = 0; k < 9; ++k) A9[i][k] = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int idx = live ? samples[(size_t)h * 5 + i] : 0;
            idx = max(min(idx, nb_pts - 1), 0);
            const double x1 = pts[(size_t)idx * 4], y1 = pts[(size_t)idx * 4 + 1];
            const double x2 = pts[(size_t)idx * 4 + 2], y2 = pts[(size_t)idx * 4 + 3];
            A9[i][0] = x2 * x1; A9[i][1] = x2 * y1; A9[i][2] = x2;
            A9[i][3] = y2 * x1; A9[i][4] = y2 * y1; A9[i][5] = y2;
            A9[i][6] = x1; A9[i][7] = y1; A9[i][8] = 1.0;
        }
        jacobi_rows_reg<9, 5, 9>(A9, W9);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int k = 0; k < 9; ++k) Nb[v][k] = A9[5 + v][k];
    }

    // the ten cubic constraints over the 20 monomials: generated straight-line code (vo_fivept_gen.h) in the operation
    // order of the generic polynomial routines kept above for reference
    double A[10][20];
    fivept_constraints(Nb, A);
    bool singular = false;
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        double best = fabs(A[col][col]);
        for (int r = col + 1; r < 10; ++r) if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); piv = r; }
        if (best < DBL_MIN) { singular = true; break; }
        if (piv != col) for (int k = 0; k < 20; ++k) { const double t = A[col][k]; A[col][k] = A[piv][k]; A[piv][k] = t; }
        const double inv = 1.0 / A[col][col];
        for (int k = 0; k < 20; ++k) A[col][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = A[r][col];
            if (f == 0) continue;
            for (int k = 0; k < 20; ++k) A[r][k] -= f * A[col][k];
        }
    }
    int count = 0;
    double B[3][3][5], cc[11];
    for (int i = 0; i < 11; ++i) cc[i] = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 5; ++k) B[i][j][k] = 0;
    if (!singular) {
        for (int i = 0; i < 3; ++i) {
            const double* ra = A[4 + 2 * i] + 10;
            const double* rb = A[5 + 2 * i] + 10;
            B[i][0][0] = ra[2]; B[i][0][1] = ra[1] - rb[2]; B[i][0][2] = ra[0] - rb[1]; B[i][0][3] = -rb[0];
            B[i][1][0] = ra[5]; B[i][1][1] = ra[4] - rb[5]; B[i][1][2] = ra[3] - rb[4]; B[i][1][3] = -rb[3];
            B[i][2][0] = ra[9]; B[i][2][1] = ra[8] - rb[9]; B[i][2][2] = ra[7] - rb[8];
            B[i][2][3] = ra[6] - rb[7]; B[i][2][4] = -rb[6];
        }
        double t1[8], t2[8], t3[12], m[8];
        double* const c = cc;
        z_mul(B[1][1], 3, B[2][2], 4, t1); z_mul(B[1][2], 4, B[2][1], 3, t2);
        for (int i = 0; i < 8; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][0], 3, m, 7, t3);
        for (int i = 0; i < 11; ++i) c[i] += t3[i];
        z_mul(B[1][0], 3, B[2][2], 4, t1); z_mul(B[1][2], 4, B[2][0], 3, t2);
        for (int i = 0; i < 8; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][1], 3, m, 7, t3);
        for (int i = 0; i < 11; ++i) c[i] -= t3[i];
        z_mul(B[1][0], 3, B[2][1], 3, t1); z_mul(B[1][1], 3, B[2][0], 3, t2);
        for (int i = 0; i < 7; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][2], 4, m, 6, t3);
        for (int i = 0; i < 11; ++i) c[i] += t3[i];

    }
    // degree-10 polynomial in z: roots by Durand-Kerner.  All lanes of the wave take the register-resident form when
    // every one of them has a full-degree polynomial (the normal case); otherwise the generic one.
    double c[11], re[10], im[10];
#pragma unroll
    for (int i = 0; i < 11; ++i) c[i] = singular ? (i == 10 ? 1. : 0.) : cc[i];
    int nr = 10;
    if (__all(fabs(c[10]) > DBL_EPSILON)) {
        solve_poly_fixed<10>(c, re, im);
    } else {
        double cm[11], rm[10], imm[10];
        for (int i = 0; i < 11; ++i) cm[i] = c[i];
        nr = solve_poly(cm, 10, rm, imm);
#pragma unroll
        for (int i = 0; i < 10; ++i) { re[i] = rm[i]; im[i] = imm[i]; }
    }
    if (!singular) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const bool real_root = i < nr && count < 10 && !(fabs(im[i]) > 1e-10);
            if (!__any(real_root)) continue;                 // wave-uniform
            const double z1 = re[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
            double At[3][3], v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                At[0][j] = B[j][0][3] * z3 + B[j][0][2] * z2 + B[j][0][1] * z1 + B[j][0][0];
                At[1][j] = B[j][1][3] * z3 + B[j][1][2] * z2 + B[j][1][1] * z1 + B[j][1][0];
                At[2][j] = B[j][2][4] * z4 + B[j][2][3] * z3 + B[j][2][2] * z2 + B[j][2][1] * z1 + B[j][2][0];
            }
            jacobi3_last_vt(At, v);
            if (!real_root || fabs(v[2]) < 1e-10) continue;
            const double x = v[0] / v[2], y = v[1] / v[2];
            double Ev[9], nrm = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                Ev[k] = Nb[0][k] * x + Nb[1][k] * y + Nb[2][k] * z1 + Nb[3][k];
                nrm += Ev[k] * Ev[k];
            }
            nrm = sqrt(nrm);
            if (live)
                for (int k = 0; k < 9; ++k) models[((size_t)h * 10 + count) * 9 + k] = Ev[k] / nrm;
            count++;
        }
    }
    append_slots(list, nlist, b, h - b * per, per, 10, live ? count : 0);
    if (live) nmodels[h] = count;
}

}  // namespace

// internal launchers shared with the fused batched pipelines in vo_ransac.hip.  H = total samples
// (all problems), per = samples per problem, N = correspondences per problem (array stride).
int vo_launch_solve_7pt(const int32_t* samples, int H, int per, const float* pts, int N, const int32_t* n_dev,
                        double* models, int32_t* nmodels, int32_t* list, int32_t* nlist, int h0, int hlen,
                        const int32_t* prev, hipStream_t s)
{
    if (hlen <= 0) return VO_OK;
    hipLaunchKernelGGL(solve_7pt_kernel, dim3(vo_ceil_div((H / per) * hlen, kLanes)), dim3(kLanes), 0, s, samples, H,
                       (const float4*)pts, N, models, nmodels, per, list, nlist, n_dev, h0, hlen, prev);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_solve_5pt(const int32_t* samples, int H, int per, const double* pts, int N, const int32_t* n_dev,
                        double* models, int32_t* nmodels, int32_t* list, int32_t* nlist, int h0, int hlen,
                        const int32_t* prev, hipStream_t s)
{
    if (hlen <= 0) return VO_OK;
    hipLaunchKernelGGL(solve_5pt_kernel, dim3(vo_ceil_div((H / per) * hlen, kLanes)), dim3(kLanes), 0, s, samples, H, pts, N,
                       models, nmodels, per, list, nlist, n_dev, h0, hlen, prev);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_solve_p3p(const int32_t* samples, int H, int per, const float* obj, const float* img, const double* K4,
                        int N, const int32_t* n_dev, double* models, double* rvecs, int32_t* nmodels, int32_t* list,
                        int32_t* nlist, int h0, int hlen, const int32_t* prev, hipStream_t s)
{
    if (hlen <= 0) return VO_OK;
    hipLaunchKernelGGL(solve_p3p_kernel, dim3(vo_ceil_div((H / per) * hlen, kLanes)), dim3(kLanes), 0, s, samples, H, obj,
                       (const float2*)img, N, K4[0], K4[1], K4[2], K4[3], models, rvecs, nmodels, per, list, nlist, n_dev,
                       h0, hlen, prev);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

#define SOLVE_ARG_CHECKS(name)                                                             \
    VO_CHECK_ARG(H >= 0 && N >= 0, VO_EINVAL, name ": negative size");                     \
    if (H == 0) return VO_OK;                                                              \
    VO_CHECK_ARG(N > 0, VO_EINVAL, name ": no correspondences");                           \
    VO_CHECK_ARG(samples && models && nmodels, VO_EINVAL, name ": null pointer")

extern "C" int vo_solve_7pt(const int32_t* samples, int H, const float* pts, int N, double* models,
                            int32_t* nmodels, void* stream)
{
    SOLVE_ARG_CHECKS("vo_solve_7pt");
    VO_CHECK_ARG(pts && ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_solve_7pt: pts null or not 16-byte aligned");
    return vo_launch_solve_7pt(samples, H, H, pts, N, nullptr, models, nmodels, nullptr, nullptr, 0, H, nullptr, (hipStream_t)stream);
}

extern "C" int vo_solve_5pt(const int32_t* samples, int H, const double* pts, int N, double* models,
                            int32_t* nmodels, void* stream)
{
    SOLVE_ARG_CHECKS("vo_solve_5pt");
    VO_CHECK_ARG(pts != nullptr, VO_EINVAL, "vo_solve_5pt: null pts");
    return vo_launch_solve_5pt(samples, H, H, pts, N, nullptr, models, nmodels, nullptr, nullptr, 0, H, nullptr, (hipStream_t)stream);
}

extern "C" int vo_solve_p3p(const int32_t* samples, int H, const float* obj, const float* img,
                            const double* K4_host, int N, double* models, double* rvecs, int32_t* nmodels,
                            void* stream)
{
    SOLVE_ARG_CHECKS("vo_solve_p3p");
    VO_CHECK_ARG(obj && img && K4_host && rvecs && ((uintptr_t)img & 7) == 0, VO_EINVAL,
                 "vo_solve_p3p: null/unaligned pointer");
    return vo_launch_solve_p3p(samples, H, H, obj, img, K4_host, N, nullptr, models, rvecs, nmodels, nullptr, nullptr,
                               0, H, nullptr, (hipStream_t)stream);
}
