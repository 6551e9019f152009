// vo_pipeline.hip -- the glue that keeps a frame pair on the device between matching and RANSAC
// (SURVEY.md 8f-1):
//   * vo_match_compact: the gathers of reference src/features.py:110-122 and src/twoDtwoD.py:54-55
//     (matched keypoint positions in ascending query order), plus the (2D, 3D) pairing that
//     src/sift_test.py:104-119 builds for pnp, as one ordered stream compaction per pair;
//   * vo_gen_subsets_device: RANSACPointSetRegistrator::getSubset driven by cv::RNG((uint64)-1)
//     (OpenCV calib3d ptsetreg.cpp) for a per-pair correspondence count that
//     only exists on the device.  The generator is sequential, but multiply-with-carry has a closed
//     form: its 64-bit state obeys s' = a*s mod (a*2^32 - 1), so the state k draws ahead is a^k * s.
//     A wave therefore evaluates 64 iterations at once under the assumption that every iteration
//     before it used exactly `model_points` draws, accepts the iterations up to the first one that
//     used more (a duplicate index or a rejected sample), and restarts behind it.
#include "vo_common.h"
#include <float.h>

namespace {

constexpr int kCompactThreads = 1024;

__global__ __launch_bounds__(kCompactThreads)
void match_compact_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ keep, int nq, int nt,
                          const float2* __restrict__ kp_q, const float2* __restrict__ kp_t,
                          const float* __restrict__ xyz_q, int32_t* __restrict__ match_q,
                          int32_t* __restrict__ match_t, int32_t* __restrict__ n_match,
                          float4* __restrict__ pts4, float* __restrict__ obj, float2* __restrict__ img)
{
    __shared__ int s_wcnt[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    idx += (size_t)b * nq * 2; keep += (size_t)b * nq;
    kp_q += (size_t)b * nq; kp_t += (size_t)b * nt;
    if (xyz_q) xyz_q += (size_t)b * nq * 3;
    match_q += (size_t)b * nq; match_t += (size_t)b * nq;
    if (pts4) pts4 += (size_t)b * nq;
    if (obj) obj += (size_t)b * nq * 3;
    if (img) img += (size_t)b * nq;
    int base = 0;
    for (int i0 = 0; i0 < nq; i0 += kCompactThreads) {
        const int i = i0 + tid;
        int t = -1;
        if (i < nq && keep[i]) t = idx[2 * i];
        const bool k = t >= 0 && t < nt;
        const unsigned long long bal = __ballot(k);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base + __popcll(bal & ((1ull << lane) - 1)), total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = s_wcnt[w];
            if (w < wave) off += c;
            total += c;
        }
        if (k) {
            match_q[off] = i;
            match_t[off] = t;
            const float2 a = kp_q[i], c = kp_t[t];
            if (pts4) pts4[off] = make_float4(a.x, a.y, c.x, c.y);
            if (img) img[off] = c;
            if (obj && xyz_q) { obj[3 * off] = xyz_q[3 * i]; obj[3 * off + 1] = xyz_q[3 * i + 1]; obj[3 * off + 2] = xyz_q[3 * i + 2]; }
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) n_match[b] = base;
}

// ---- cv::RNG as a Lehmer generator modulo m = a * 2^32 - 1 -------------------------------------
constexpr unsigned long long kA = 4164903690ull;
constexpr unsigned long long kM = kA * 4294967296ull - 1ull;

__host__ __device__ inline unsigned long long rng_step(unsigned long long s)
{
    return (unsigned long long)(unsigned)s * kA + (s >> 32);
}

__host__ __device__ inline unsigned long long addmod(unsigned long long x, unsigned long long y)
{
    return x >= kM - y ? x - (kM - y) : x + y;          // x, y < m < 2^64
}

__host__ __device__ inline unsigned long long mulmod(unsigned long long x, unsigned long long y)
{
    unsigned long long r = 0;
    for (int bit = 63; bit >= 0; --bit) {
        r = addmod(r, r);
        if ((y >> bit) & 1) r = addmod(r, x);
    }
    return r;
}

// x * y * 2^-64 mod m, x, y < m.  a * 2^32 = 1 (mod m), so for any X:  X = 2^32 (X >> 32) + lo32(X)
// = 2^32 ((X >> 32) + lo32(X) a)  (mod m): one multiply-by-a step (what rng_step does to a state) divides by 2^32.
// Two of them bring the 128-bit product down to 65 bits; the factor 2^-64 is folded into the jump table on the host.
__device__ inline unsigned long long mulmod_redc(unsigned long long x, unsigned long long y)
{
    unsigned __int128 p = (unsigned __int128)x * y;
    p = (p >> 32) + (unsigned __int128)(unsigned)p * kA;       // < 2^96 + 2^64
    p = (p >> 32) + (unsigned __int128)(unsigned)p * kA;       // < 2^64 + 2^64
    while (p >= kM) p -= kM;                                    // at most three times
    return (unsigned long long)p;
}

constexpr int kSubsetThreads = 256;   // speculative samples per round and problem (4 waves)

struct SubsetArgs {
    const int32_t* n_dev; const float4* pts4; int32_t* samples; int32_t* n_generated;
    int N, model_points, H, check, max_attempts;
    unsigned long long jump[kSubsetThreads + 1];      // a^(model_points * l) * 2^64 mod m (see mulmod_redc)
};

// last point of the sample (near-)collinear with a pair of earlier ones, in image `off` (0 or 2)
template <int COUNT>
__device__ inline bool collinear_last(const float4* pts, const int (&s)[COUNT], int off)
{
    constexpr int count = COUNT;
    const float* p = (const float*)pts;
    const int i = count - 1;
    const float xi = p[4 * s[i] + off], yi = p[4 * s[i] + off + 1];
#pragma unroll
    for (int j = 0; j < i; ++j) {
        const double dx1 = (double)(p[4 * s[j] + off] - xi), dy1 = (double)(p[4 * s[j] + off + 1] - yi);
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)(p[4 * s[k] + off] - xi), dy2 = (double)(p[4 * s[k] + off + 1] - yi);
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

template <int M>
__global__ __launch_bounds__(kSubsetThreads)
void gen_subsets_kernel(const SubsetArgs a)
{
    constexpr int m = M;                                  // compile-time: the sample lives in registers
    __shared__ unsigned long long s_state[kSubsetThreads];
    __shared__ unsigned long long s_odd[kSubsetThreads / 64], s_act[kSubsetThreads / 64];
    __shared__ unsigned char s_fail[kSubsetThreads];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    int32_t* out = a.samples + (size_t)b * a.H * m;
    const float4* pts = a.pts4 ? a.pts4 + (size_t)b * a.N : nullptr;
    int done = 0;                                         // iterations generated so far (block-uniform)
    bool failed = n < m;
    // two plain steps bring the seed 0xFFFF...F into the canonical range [0, m) of the closed form;
    // they are real draws, so they are replayed below as the first two draws of iteration 0
    unsigned long long base = 0xFFFFFFFFFFFFFFFFull;
    int warm = 2;                                         // draws still to be taken sequentially
    while (!failed && done < a.H) {
        unsigned long long s;
        if (warm > 0) {
            // the first chunk is one iteration wide: thread 0 walks the generator step by step
            s = base;
        } else {
            s = mulmod_redc(a.jump[tid], base);
        }
        const bool active = warm > 0 ? tid == 0 : done + tid < a.H;
        int sub[M];
        int used = 0;
        bool ok = false;
        if (active) {
            for (int attempt = 0; attempt < a.max_attempts && !ok; ++attempt) {
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    int v;
                    for (;;) {
                        s = rng_step(s);
                        ++used;
                        v = (int)((unsigned)s % (unsigned)n);
                        bool dup = false;
#pragma unroll
                        for (int k = 0; k < M; ++k) dup |= k < i && sub[k] == v;
                        if (!dup) break;
                    }
                    sub[i] = v;
                }
                ok = !(a.check && pts) || (!collinear_last(pts, sub, 0) && !collinear_last(pts, sub, 2));
            }
        }
        // threads up to the first one that did not use exactly m draws started from the right state
        const unsigned long long odd = __ballot(active && (used != m || !ok));
        const unsigned long long act = __ballot(active);
        if ((tid & 63) == 0) { s_odd[wave] = odd; s_act[wave] = act; }
        s_state[tid] = s;
        s_fail[tid] = !ok;
        __syncthreads();
        int nact = 0, first_odd = -1;
#pragma unroll
        for (int w = 0; w < kSubsetThreads / 64; ++w) {
            const unsigned long long o = s_odd[w];
            if (first_odd < 0 && o) first_odd = 64 * w + __ffsll((long long)o) - 1;
            nact += __popcll(s_act[w]);
        }
        if (first_odd < 0) first_odd = nact - 1;
        const bool bad = s_fail[first_odd] != 0;          // that thread exhausted its attempts
        const int nwrite = bad ? first_odd : first_odd + 1;
        if (active && tid < nwrite) {
#pragma unroll
            for (int i = 0; i < m; ++i) out[(size_t)(done + tid) * m + i] = sub[i];
        }
        done += nwrite;
        base = s_state[first_odd];                        // generator state behind the last valid thread
        warm = 0;
        __syncthreads();                                  // the shared arrays are rewritten next round
        if (bad) failed = true;
    }
    // iterations that were never generated: mark "no sample" for the solvers
    for (long i = (long)done * m + tid; i < (long)a.H * m; i += kSubsetThreads) out[i] = -1;
    if (tid == 0 && a.n_generated) a.n_generated[b] = done;
}

}  // namespace

extern "C" int vo_match_compact(const int32_t* idx, const uint8_t* keep, int nq, int nt, int batch,
                                const float* kp_q, const float* kp_t, const float* xyz_q,
                                int32_t* match_q, int32_t* match_t, int32_t* n_match,
                                float* pts4, float* obj, float* img, void* stream)
{
    VO_CHECK_ARG(nq >= 0 && nt >= 0 && batch >= 0 && batch <= 65535, VO_EINVAL, "vo_match_compact: bad sizes");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(idx && keep && kp_q && kp_t && match_q && match_t && n_match, VO_EINVAL, "vo_match_compact: null pointer");
    VO_CHECK_ARG(!obj || xyz_q, VO_EINVAL, "vo_match_compact: obj output needs xyz_q");
    VO_CHECK_ARG((((uintptr_t)kp_q | (uintptr_t)kp_t | (uintptr_t)img) & 7) == 0 && ((uintptr_t)pts4 & 15) == 0, VO_EINVAL,
                 "vo_match_compact: keypoint arrays must be 8-byte, pts4 16-byte aligned");
    hipLaunchKernelGGL(match_compact_kernel, dim3(batch), dim3(kCompactThreads), 0, (hipStream_t)stream, idx, keep, nq, nt,
                       (const float2*)kp_q, (const float2*)kp_t, xyz_q, match_q, match_t, n_match, (float4*)pts4, obj,
                       (float2*)img);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

int vo_launch_gen_subsets(const int32_t* n_dev, int N, int model_points, int H, const float* pts4, int batch,
                          int32_t* samples, int32_t* n_generated, int max_attempts, hipStream_t s)
{
    SubsetArgs a;
    a.n_dev = n_dev; a.pts4 = (const float4*)pts4; a.samples = samples; a.n_generated = n_generated;
    a.N = N; a.model_points = model_points; a.H = H; a.check = pts4 != nullptr; a.max_attempts = max_attempts;
    // a^(model_points * l) mod m by repeated multiplication with a^model_points
    unsigned long long am = 1;
    for (int i = 0; i < model_points; ++i) am = mulmod(am, kA);
    a.jump[0] = 0ull - kM;                              // 2^64 mod m (m < 2^64 < 2 m)
    for (int l = 1; l <= kSubsetThreads; ++l) a.jump[l] = mulmod(a.jump[l - 1], am);
    if (model_points == 4) hipLaunchKernelGGL(gen_subsets_kernel<4>, dim3(batch), dim3(kSubsetThreads), 0, s, a);
    else if (model_points == 5) hipLaunchKernelGGL(gen_subsets_kernel<5>, dim3(batch), dim3(kSubsetThreads), 0, s, a);
    else hipLaunchKernelGGL(gen_subsets_kernel<7>, dim3(batch), dim3(kSubsetThreads), 0, s, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_gen_subsets_device(const int32_t* n_dev, int N, int model_points, int H, const float* pts4,
                                     int batch, int32_t* samples, int32_t* n_generated, void* stream)
{
    VO_CHECK_ARG((model_points == 4 || model_points == 5 || model_points == 7) && H >= 0 && N >= 0 && batch >= 0 &&
                 batch <= 65535, VO_EINVAL, "vo_gen_subsets_device: bad sizes (model_points must be 4, 5 or 7)");
    if (batch == 0 || H == 0) return VO_OK;
    VO_CHECK_ARG(samples != nullptr && ((uintptr_t)pts4 & 15) == 0, VO_EINVAL, "vo_gen_subsets_device: null/unaligned pointer");
    return vo_launch_gen_subsets(n_dev, N, model_points, H, pts4, batch, samples, n_generated, 10000, (hipStream_t)stream);
}
