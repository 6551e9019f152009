// vo_score.hip -- RANSAC hypothesis scoring on gfx950: for H models x N correspondences,
// the error OpenCV's computeError callbacks define and the inlier count findInliers derives
// from it.  Stands in for the inner loop of cv2.findFundamentalMat / cv2.findEssentialMat /
// cv.solvePnPRansac (reference src/utils/geometry.py:39,44 and src/pnp.py:16-26; semantics:
// SURVEY.md Appendix A.2-A.6, Appendix B for the exactness rules).
//
// Exactness: every value that reaches the threshold compare is produced by the same IEEE
// f64/f32 operations in the same order as the scalar definition (this file is compiled with
// -ffp-contract=off; f64 divide and f32->f64->f32 conversions are correctly rounded on
// gfx950), so counts and masks are bit-for-bit those of a CPU evaluation.
//
// Mapping: one thread owns one model (9 or 12 doubles in VGPRs); a workgroup of 256 models
// walks a chunk of correspondences that it first converts to f64 in LDS, so the inner loop
// reads every operand with wave-uniform (broadcast) ds_read_b128 and is pure f64 VALU.  The
// grid is (model blocks) x (point chunks); chunk sums meet in counts[] through integer
// atomics, which are order-independent.
#include "vo_common.h"

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

__global__ void init_counts_kernel(const uint8_t* __restrict__ valid, int H, int32_t* __restrict__ counts)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < H) counts[h] = (valid && !valid[h]) ? -1 : 0;
}

// ---- count kernels ------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads)
void score_fund_kernel(const double* __restrict__ models, const uint8_t* __restrict__ valid, int H,
                       const float4* __restrict__ pts, int N, int chunk, float thr2,
                       int32_t* __restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) double sp[kMaxChunk * 4];
    const int h = blockIdx.x * kThreads + threadIdx.x;
    const int p0 = blockIdx.y * chunk, np = min(chunk, N - p0);
    for (int i = threadIdx.x; i < np; i += kThreads) {
        const float4 v = pts[p0 + i];
        sp[4 * i] = v.x; sp[4 * i + 1] = v.y; sp[4 * i + 2] = v.z; sp[4 * i + 3] = v.w;
    }
    const bool live = h < H && (!valid || valid[h]);
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = live ? models[(size_t)h * 9 + k] : 0.;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < np; ++i) {
        const double2 a = *(const double2*)(sp + 4 * i), b = *(const double2*)(sp + 4 * i + 2);
        cnt += fund_err(F, a.x, a.y, b.x, b.y) <= thr2;
    }
    if (live && cnt) atomicAdd(counts + h, cnt);
}

__global__ __launch_bounds__(kThreads)
void score_ess_kernel(const double* __restrict__ models, const uint8_t* __restrict__ valid, int H,
                      const double* __restrict__ pts, int N, int chunk, float thr2,
                      int32_t* __restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) double sp[kMaxChunk * 4];
    const int h = blockIdx.x * kThreads + threadIdx.x;
    const int p0 = blockIdx.y * chunk, np = min(chunk, N - p0);
    for (int i = threadIdx.x; i < np * 4; i += kThreads) sp[i] = pts[(size_t)p0 * 4 + i];
    const bool live = h < H && (!valid || valid[h]);
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = live ? models[(size_t)h * 9 + k] : 0.;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < np; ++i) {
        const double2 a = *(const double2*)(sp + 4 * i), b = *(const double2*)(sp + 4 * i + 2);
        cnt += sampson_err(E, a.x, a.y, b.x, b.y) <= thr2;
    }
    if (live && cnt) atomicAdd(counts + h, cnt);
}

__global__ __launch_bounds__(kThreads)
void score_pnp_kernel(const double* __restrict__ models, const uint8_t* __restrict__ valid, int H,
                      const float* __restrict__ obj, const float2* __restrict__ img, int N, int chunk,
                      double fx, double fy, double cx, double cy, float thr2,
                      int32_t* __restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) double sp[kMaxChunk * 4];   // X Y Z (u,v as 2 x f32)
    const int h = blockIdx.x * kThreads + threadIdx.x;
    const int p0 = blockIdx.y * chunk, np = min(chunk, N - p0);
    for (int i = threadIdx.x; i < np; i += kThreads) {
        const float* o = obj + (size_t)(p0 + i) * 3;
        sp[4 * i] = o[0]; sp[4 * i + 1] = o[1]; sp[4 * i + 2] = o[2];
        *(float2*)(sp + 4 * i + 3) = img[p0 + i];
    }
    const bool live = h < H && (!valid || valid[h]);
    double M[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) M[k] = live ? models[(size_t)h * 12 + k] : 0.;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < np; ++i) {
        const double2 a = *(const double2*)(sp + 4 * i);
        const double Z = sp[4 * i + 2];
        const float2 uv = *(const float2*)(sp + 4 * i + 3);
        cnt += pnp_err(M, fx, fy, cx, cy, a.x, a.y, Z, uv.x, uv.y) <= thr2;
    }
    if (live && cnt) atomicAdd(counts + h, cnt);
}

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

int pick_chunk(int H, int N)
{
    const int mblocks = vo_ceil_div(H, kThreads);
    const int target = 2048;                      // ~8 workgroups of 4 waves per CU
    int per = vo_ceil_div(target, mblocks);       // chunks wanted along N
    int chunk = vo_ceil_div(N, per);
    chunk = vo_ceil_div(chunk, 32) * 32;
    if (chunk < 32) chunk = 32;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    return chunk;
}

int launch_init(const uint8_t* valid, int H, int32_t* counts, hipStream_t s)
{
    hipLaunchKernelGGL(init_counts_kernel, dim3(vo_ceil_div(H, 256)), dim3(256), 0, s, valid, H, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

}  // namespace

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
    int rc = launch_init(valid, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(pts && ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_score_fund: pts null or not 16-byte aligned");
    const int chunk = pick_chunk(H, N);
    VO_CHECK_ARG(vo_ceil_div(N, chunk) <= 65535, VO_ESIZE, "vo_score_fund: N=%d too large", N);
    hipLaunchKernelGGL(score_fund_kernel, dim3(vo_ceil_div(H, kThreads), vo_ceil_div(N, chunk)), dim3(kThreads),
                       0, s, models, valid, H, (const float4*)pts, N, chunk, thr2, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_score_ess(const double* models, const uint8_t* valid, int H, const double* pts, int N,
                            float thr2, int32_t* counts, void* stream)
{
    SCORE_ARG_CHECKS("vo_score_ess");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_init(valid, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(pts != nullptr, VO_EINVAL, "vo_score_ess: null pts");
    const int chunk = pick_chunk(H, N);
    VO_CHECK_ARG(vo_ceil_div(N, chunk) <= 65535, VO_ESIZE, "vo_score_ess: N=%d too large", N);
    hipLaunchKernelGGL(score_ess_kernel, dim3(vo_ceil_div(H, kThreads), vo_ceil_div(N, chunk)), dim3(kThreads),
                       0, s, models, valid, H, pts, N, chunk, thr2, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_score_pnp(const double* models, const uint8_t* valid, int H, const float* obj,
                            const float* img, const double* K4_host, int N, float thr2, int32_t* counts,
                            void* stream)
{
    SCORE_ARG_CHECKS("vo_score_pnp");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_init(valid, H, counts, s);
    if (rc != VO_OK || N == 0) return rc;
    VO_CHECK_ARG(obj && img && K4_host, VO_EINVAL, "vo_score_pnp: null pointer");
    VO_CHECK_ARG(((uintptr_t)img & 7) == 0, VO_EINVAL, "vo_score_pnp: img must be 8-byte aligned");
    const int chunk = pick_chunk(H, N);
    VO_CHECK_ARG(vo_ceil_div(N, chunk) <= 65535, VO_ESIZE, "vo_score_pnp: N=%d too large", N);
    hipLaunchKernelGGL(score_pnp_kernel, dim3(vo_ceil_div(H, kThreads), vo_ceil_div(N, chunk)), dim3(kThreads),
                       0, s, models, valid, H, obj, (const float2*)img, N, chunk, K4_host[0], K4_host[1],
                       K4_host[2], K4_host[3], thr2, counts);
    VO_LAUNCH_CHECK();
    return VO_OK;
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
