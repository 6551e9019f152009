// vo_pose.hip -- the O(N) tail of the 2D-2D initialisation on the device:
//   * cv.triangulatePoints (reference src/utils/disambiguate_relative_pose.py:42,
//     src/twoDtwoD.py:110, src/tracking.py:153): per correspondence the 4x4 DLT system, solved by the
//     right singular vector of the smallest singular value (one-sided Jacobi, the operation order
//     of the oracle's restatement of OpenCV's JacobiSVD, so the homogeneous points agree bit for bit);
//   * E = K^T F K, decomposeEssentialMatrix and disambiguateRelativePose
//     (reference src/utils/geometry.py:48-49, src/utils/decompose_essential_matrix.py:4-37,
//     src/utils/disambiguate_relative_pose.py:5-72) for a batch of frame pairs straight from the
//     RANSAC results, without a host round trip.
// The 3x3 SVD of E is a Jacobi SVD here and LAPACK in the reference: singular vectors agree up to
// the sign freedom of the decomposition, which can permute the order of the four (R, +-t)
// candidates; the candidate SET is the same and the winner is the same whenever the cheirality
// score has a unique maximum (any non-degenerate scene).  Poses agree to rounding, not bit for bit.
#include "vo_common.h"
#include <float.h>

namespace {

// Right singular vector of the smallest singular value of the 4x4 matrix whose COLUMNS are the rows
// of At (At = A^T for the DLT matrix A).  One-sided Jacobi on the rows of At with V accumulated;
// rows sorted by descending norm; returns Vt[3].  All indices compile-time: registers only.
__device__ __forceinline__ void svd4_last_vt(double (&At)[4][4], double (&out)[4])
{
    const double eps = DBL_EPSILON * 10;
    double W[4], Vt[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < 4; ++k) Vt[i][k] = i == k ? 1. : 0.;
    }
#pragma unroll 1
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) p += At[i][k] * At[j][k];
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
                    for (int k = 0; k < 4; ++k) {
                        const double t0 = c * At[i][k] + s * At[j][k];
                        const double t1 = -s * At[i][k] + c * At[j][k];
                        At[i][k] = t0; At[j][k] = t1;
                        a += t0 * t0; b += t1 * t1;
                    }
                    W[i] = a; W[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double t0 = c * Vt[i][k] + s * Vt[j][k];
                        const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                        Vt[i][k] = t0; Vt[j][k] = t1;
                    }
                }
            }
        if (!__any(changed)) break;       // converged lanes ride along: nothing left to rotate
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < 4; ++k)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < 4; ++k)
            if (j == k) {
                double t = W[i]; W[i] = W[k]; W[k] = t;
#pragma unroll
                for (int e = 0; e < 4; ++e) { t = Vt[i][e]; Vt[i][e] = Vt[k][e]; Vt[k][e] = t; }
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = Vt[3][k];
}

// cv.triangulatePoints for one correspondence: rows  x P[2]-P[0],  y P[2]-P[1]  of either camera
__device__ __forceinline__ void triangulate_one(const double* P1, const double* P2, double x1, double y1,
                                                double x2, double y2, double (&X)[4])
{
    double At[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        At[k][0] = x1 * P1[8 + k] - P1[k];
        At[k][1] = y1 * P1[8 + k] - P1[4 + k];
        At[k][2] = x2 * P2[8 + k] - P2[k];
        At[k][3] = y2 * P2[8 + k] - P2[4 + k];
    }
    svd4_last_vt(At, X);
}

struct Proj { double p[12]; };

__global__ __launch_bounds__(256)
void triangulate_kernel(Proj P1, Proj P2, const double* __restrict__ x1, const double* __restrict__ x2, int N,
                        double* __restrict__ X4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    if (!__any(live)) return;
    const int ii = live ? i : 0;
    double X[4];
    triangulate_one(P1.p, P2.p, x1[2 * ii], x1[2 * ii + 1], x2[2 * ii], x2[2 * ii + 1], X);
    if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) X4[(size_t)4 * i + k] = X[k];
    }
}

// ---- batched pose tail --------------------------------------------------------------------------

__device__ void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Jacobi SVD of a 3x3 matrix of rank >= 2 -> U, V with det +1: the two leading singular pairs from
// the rotation sweeps, the third vectors as cross products (E has a vanishing third singular value,
// whose Jacobi row is rounding noise).
__device__ void svd3_rank2(const double* E, double U[3][3], double V[3][3])
{
    const double eps = DBL_EPSILON * 10;
    double At[3][3], Vt[3][3], W[3];
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 3; ++k) { At[i][k] = E[k * 3 + i]; Vt[i][k] = i == k ? 1. : 0.; }
        W[i] = At[i][0] * At[i][0] + At[i][1] * At[i][1] + At[i][2] * At[i][2];
    }
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 3; ++k) p += At[i][k] * At[j][k];
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
                for (int k = 0; k < 3; ++k) {
                    const double t0 = c * At[i][k] + s * At[j][k], t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; b += t1 * t1;
                    const double v0 = c * Vt[i][k] + s * Vt[j][k], v1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = v0; Vt[j][k] = v1;
                }
                W[i] = a; W[j] = b;
                changed = true;
            }
        if (!changed) break;
    }
    // the two largest rows
    int o[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int k = i + 1; k < 3; ++k)
            if (W[o[i]] < W[o[k]]) { const int t = o[i]; o[i] = o[k]; o[k] = t; }
    double u[2][3], v[2][3];
    for (int r = 0; r < 2; ++r) {
        const double nrm = sqrt(W[o[r]]);
        const double s = nrm > 0 ? 1. / nrm : 0.;
        for (int k = 0; k < 3; ++k) { u[r][k] = At[o[r]][k] * s; v[r][k] = Vt[o[r]][k]; }
    }
    double u2[3], v2[3];
    cross3(u[0], u[1], u2);
    cross3(v[0], v[1], v2);
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (n2 > 0) for (int k = 0; k < 3; ++k) u2[k] /= n2;
    for (int k = 0; k < 3; ++k) {
        U[k][0] = u[0][k]; U[k][1] = u[1][k]; U[k][2] = u2[k];
        V[k][0] = v[0][k]; V[k][1] = v[1][k]; V[k][2] = v2[k];
    }
}

struct TailArgs {
    const double* F;          // batch x 9
    const int32_t* result;    // batch x 4 (vo_ransac_batch); NULL = every pair has a model
    const uint8_t* mask;      // batch x N inliers of F
    const float4* pts4;       // batch x N
    const int32_t* n_dev;     // batch, may be NULL
    double* cand;             // batch x 4 x 12: the four [R|t] candidates, reference order
    int32_t* score;           // batch x 4
    double* T;                // batch x 16
    double* X;                // batch x N x 4
    uint8_t* keep;            // batch x N
    double K[9];
    double max_depth;
    int N;
};

__device__ __forceinline__ bool pair_has_model(const TailArgs& a, int b) { return !a.result || a.result[4 * b] >= 0; }

// one thread per pair: E = K^T F K, decomposeEssentialMatrix -> four candidates; clears the scores
__global__ void tail_candidates_kernel(const TailArgs a, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    for (int c = 0; c < 4; ++c) a.score[4 * b + c] = 0;
    double* cand = a.cand + (size_t)b * 48;
    if (!pair_has_model(a, b)) {
        for (int k = 0; k < 48; ++k) cand[k] = 0;
        return;
    }
    const double* F = a.F + (size_t)b * 9;
    double FK[9], E[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += F[i * 3 + k] * a.K[k * 3 + j];
            FK[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a.K[k * 3 + i] * FK[k * 3 + j];
            E[i * 3 + j] = s;
        }
    double U[3][3], V[3][3];
    svd3_rank2(E, U, V);
    // R0 = U W V^T = u1 v0^T - u0 v1^T + u2 v2^T ; R1 = U W^T V^T = -u1 v0^T + u0 v1^T + u2 v2^T
    for (int r = 0; r < 2; ++r) {
        const double sg = r == 0 ? 1. : -1.;
        for (int sign = 0; sign < 2; ++sign) {
            double* M = cand + (size_t)(2 * r + sign) * 12;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j)
                    M[i * 4 + j] = sg * (U[i][1] * V[j][0] - U[i][0] * V[j][1]) + U[i][2] * V[j][2];
                M[i * 4 + 3] = sign == 0 ? U[i][2] : -U[i][2];
            }
        }
    }
}

__device__ __forceinline__ void k_times_rt(const double* K, const double* M, double* P)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * M[k * 4 + j];
            P[i * 4 + j] = s;
        }
}

// Both kernels below work on chunks of kTailChunk correspondences of one pair (grid: chunks x pairs).  Only the F
// inliers (typically 15-40 % of the matches, scattered) are triangulated, so each workgroup first gathers the
// indices of its chunk's inliers into LDS and then walks that list with full waves; a wave in which one lane in six
// has work would cost as much as a full one (the Jacobi sweeps are wave-uniform loops).
constexpr int kTailChunk = 2048, kTailThreads = 256;

__device__ __forceinline__ int tail_gather_inliers(const TailArgs& a, int b, int n, int base, int* s_list, int* s_n)
{
    if (threadIdx.x == 0) *s_n = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < kTailChunk; k += kTailThreads) {
        const int i = base + k;
        if (i < n && a.mask[(size_t)b * a.N + i] != 0) s_list[atomicAdd(s_n, 1)] = i;     // order is irrelevant
    }
    __syncthreads();
    return *s_n;
}

// cheirality score of the four candidates
__global__ __launch_bounds__(kTailThreads)
void tail_score_kernel(const TailArgs a)
{
    __shared__ double sP[4][12], sM[4][12], sP1[12];
    __shared__ int s_list[kTailChunk], s_n;
    const int b = blockIdx.y, base = blockIdx.x * kTailChunk;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    if (base >= n || !pair_has_model(a, b)) return;
    if (threadIdx.x < 4) {
        const double* M = a.cand + ((size_t)b * 4 + threadIdx.x) * 12;
        for (int k = 0; k < 12; ++k) sM[threadIdx.x][k] = M[k];
        k_times_rt(a.K, M, sP[threadIdx.x]);
    }
    if (threadIdx.x == 4)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) sP1[i * 4 + j] = j < 3 ? a.K[i * 3 + j] : 0.;
    const int cnt = tail_gather_inliers(a, b, n, base, s_list, &s_n);      // barriers inside also publish sP / sM
    // Candidates 2r and 2r+1 share the rotation and differ in the sign of t.  Negating t negates the fourth column
    // of the DLT matrix exactly (every entry is a difference of products that all change sign), and the one-sided
    // Jacobi iteration commutes with that sign flip bit for bit (rotation angles change sign with the affected inner
    // products, every intermediate is the exact negative or the same number), so the null vector comes out as
    // +-(X0, X1, X2, -X3): the dehomogenised point of (R, -t) is exactly minus that of (R, +t), and so is its depth in
    // the second camera.  One triangulation therefore scores both candidates, with the same bits as two would.
    int sc[4] = {0, 0, 0, 0};
    for (int t0 = 0; t0 < cnt; t0 += kTailThreads) {
        const int t = t0 + threadIdx.x;
        const bool live = t < cnt;
        if (!__any(live)) break;                // wave-uniform
        const float4 v = a.pts4[(size_t)b * a.N + s_list[live ? t : 0]];
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int c = 2 * r;
            double X[4];
            triangulate_one(sP1, sP[c], v.x, v.y, v.z, v.w, X);
            const double x = X[0] / X[3], y = X[1] / X[3], z = X[2] / X[3], w = X[3] / X[3];
            const double z2 = sM[c][8] * x + sM[c][9] * y + sM[c][10] * z + sM[c][11] * w;
            if (live) {
                sc[c] += (int)(z > 0) + (int)(z2 > 0);
                sc[c + 1] += (int)(-z > 0) + (int)(-z2 > 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int s = sc[c];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.score + 4 * b + c, s);
    }
}

// first strict maximum wins; points of the winner, depth/distance mask, T
__global__ __launch_bounds__(kTailThreads)
void tail_final_kernel(const TailArgs a)
{
    __shared__ double sP[12], sM[12], sP1[12];
    __shared__ int s_list[kTailChunk], s_n;
    const int b = blockIdx.y, base = blockIdx.x * kTailChunk;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const bool have = pair_has_model(a, b);
    int best = 0;
    if (have) {
        int bs = a.score[4 * b];
        for (int c = 1; c < 4; ++c)
            if (a.score[4 * b + c] > bs) { bs = a.score[4 * b + c]; best = c; }
    }
    const double* M = a.cand + ((size_t)b * 4 + best) * 12;
    if (threadIdx.x == 0) {
        for (int k = 0; k < 12; ++k) sM[k] = M[k];
        k_times_rt(a.K, M, sP);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) sP1[i * 4 + j] = j < 3 ? a.K[i * 3 + j] : 0.;
        if (blockIdx.x == 0) {
            double* T = a.T + (size_t)b * 16;
            for (int k = 0; k < 12; ++k) T[k] = have ? M[k] : (k % 5 == 0 ? 1. : 0.);
            T[12] = T[13] = T[14] = 0.; T[15] = 1.;
        }
    }
    // rows that are not inliers (or lie past the pair's count, or belong to a pair without a model): zeros
    for (int k = threadIdx.x; k < kTailChunk; k += kTailThreads) {
        const int i = base + k;
        if (i < a.N && !(have && i < n && a.mask[(size_t)b * a.N + i] != 0)) {
            double2* out = (double2*)(a.X + ((size_t)b * a.N + i) * 4);
            out[0] = make_double2(0., 0.); out[1] = make_double2(0., 0.);
            a.keep[(size_t)b * a.N + i] = 0;
        }
    }
    if (!have) return;                                                        // block-uniform
    const int cnt = tail_gather_inliers(a, b, n, base, s_list, &s_n);
    for (int t0 = 0; t0 < cnt; t0 += kTailThreads) {
        const int t = t0 + threadIdx.x;
        const bool live = t < cnt;
        if (!__any(live)) break;
        const int i = s_list[live ? t : 0];
        const float4 v = a.pts4[(size_t)b * a.N + i];
        double H[4];
        triangulate_one(sP1, sP, v.x, v.y, v.z, v.w, H);
        if (live) {
            double X[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) X[k] = H[k] / H[3];
            const double min_depth = sM[10] < 0. ? sM[10] : 0.;          // min(0, T[2,2])
            const double nrm = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]);
            double2* out = (double2*)(a.X + ((size_t)b * a.N + i) * 4);
            out[0] = make_double2(X[0], X[1]); out[1] = make_double2(X[2], X[3]);
            a.keep[(size_t)b * a.N + i] = X[2] > min_depth && nrm < a.max_depth;
        }
    }
}

}  // namespace

extern "C" int vo_triangulate(const double* P1_host, const double* P2_host, const double* x1, const double* x2,
                              int N, double* X4, void* stream)
{
    VO_CHECK_ARG(N >= 0, VO_EINVAL, "vo_triangulate: negative size");
    if (N == 0) return VO_OK;
    VO_CHECK_ARG(P1_host && P2_host && x1 && x2 && X4, VO_EINVAL, "vo_triangulate: null pointer");
    Proj p1, p2;
    for (int k = 0; k < 12; ++k) { p1.p[k] = P1_host[k]; p2.p[k] = P2_host[k]; }
    hipLaunchKernelGGL(triangulate_kernel, dim3(vo_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, p1, p2, x1,
                       x2, N, X4);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_pose_tail_batch(const double* F, const int32_t* result, const uint8_t* mask, const float* pts4,
                                  const int32_t* n_dev, const double* K_host, int N, int batch, double max_depth,
                                  double* cand, int32_t* score, double* T, double* X, uint8_t* keep, void* stream)
{
    VO_CHECK_ARG(N >= 0 && batch >= 0, VO_EINVAL, "vo_pose_tail_batch: negative size");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(F && K_host && cand && score && T, VO_EINVAL, "vo_pose_tail_batch: null pointer");
    VO_CHECK_ARG(N == 0 || (mask && pts4 && X && keep), VO_EINVAL, "vo_pose_tail_batch: null point buffers");
    VO_CHECK_ARG((((uintptr_t)pts4 | (uintptr_t)X) & 15) == 0, VO_EINVAL, "vo_pose_tail_batch: pts4 and X must be 16-byte aligned");
    VO_CHECK_ARG(batch <= 65535, VO_ESIZE, "vo_pose_tail_batch: batch too large");
    hipStream_t s = (hipStream_t)stream;
    TailArgs a;
    a.F = F; a.result = result; a.mask = mask; a.pts4 = (const float4*)pts4; a.n_dev = n_dev; a.cand = cand;
    a.score = score; a.T = T; a.X = X; a.keep = keep; a.max_depth = max_depth; a.N = N;
    for (int k = 0; k < 9; ++k) a.K[k] = K_host[k];
    hipLaunchKernelGGL(tail_candidates_kernel, dim3(vo_ceil_div(batch, 64)), dim3(64), 0, s, a, batch);
    VO_LAUNCH_CHECK();
    const dim3 grid(vo_ceil_div(N > 0 ? N : 1, kTailChunk), batch);
    if (N > 0) {
        hipLaunchKernelGGL(tail_score_kernel, grid, dim3(kTailThreads), 0, s, a);
        VO_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(tail_final_kernel, grid, dim3(kTailThreads), 0, s, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}
