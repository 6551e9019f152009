// Synthetic SIFT-like descriptor sets, generated on the device (no counterpart in the reference: there are no
// datasets and no SIFT detector here, and the offline sweep -- visual_odometry_amd/sweep.py -- was bound by its
// generator when that was a chain of elementwise torch ops).  HBM-write bound: 4 B stored per value, one pass.
//
// Every value is a pure function of (pair key, stream, element index): the hash is the one PairRng.bits of sweep.py
// evaluates with int64 tensor ops -- splitmix64's finaliser of  key ^ mix(index * GOLD + mix(stream))  -- so both
// generators draw the same uniforms; the float transforms behind them (log, cos, pow: hardware approximations here)
// differ by ~1e-6 relative between the two -- a different integer after the rounding for about one value in a
// thousand -- which is why sweep.py keeps to one of them per device type.
//
//   vo_synth_sift_like    one wave per descriptor row, two values per lane: Gamma(0.6)-like magnitudes
//                         (Wilson-Hilferty Gamma(1.6) x U^(1/0.6)), L2-normalise, clip 0.2, renormalise, x512,
//                         round, saturate to [0,255] -- the recipe of synth.sift_like
//   vo_synth_noisy_copies train[b, dst[b,m]] = saturate(round(query[b, src[b,m]] + sigma N(0,1))): the true matches
#include "vo_common.h"

namespace {

constexpr unsigned long long kC1 = 0xBF58476D1CE4E5B9ull, kC2 = 0x94D049BB133111EBull, kGold = 0x9E3779B97F4A7C15ull;

__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x = (x ^ (x >> 30)) * kC1;
    x = (x ^ (x >> 27)) * kC2;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float gamma06(unsigned long long b)
{
    const float s = 0x1p-21f;
    const float u1 = ((float)(b >> 43) + 0.5f) * s;
    const float u2 = ((float)((b >> 22) & 0x1FFFFF) + 0.5f) * s;
    const float u3 = ((float)((b >> 1) & 0x1FFFFF) + 0.5f) * s;
    // hardware transcendentals (v_log_f32, v_cos_f32, v_exp_f32: ~1e-6 relative): the generator is compute bound on
    // these, and its values are rounded to integers afterwards
    // (v_log_f32 = log2, v_cos_f32 takes revolutions, v_exp_f32 = 2^x; every argument is a normal number in (0, 1))
    const float zn = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
    const float c = 1.0f / (9.0f * 1.6f);
    const float w = fmaxf(1.0f - c + 0.26352313834736496f * zn, 0.0f);
    return 1.6f * (w * w * w) * __builtin_amdgcn_exp2f((1.0f / 0.6f) * __builtin_amdgcn_logf(u3));
}

__global__ __launch_bounds__(256)
void sift_like_kernel(const long long* __restrict__ keys, int n, long rows, unsigned long long stream_mix, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nwaves) {
        const long b = r / n, row = r - b * n;
        const unsigned long long key = (unsigned long long)keys[b];
        const unsigned long long e0 = (unsigned long long)(row * 128 + 2 * lane);
        float v0 = gamma06(mix64(key ^ mix64(e0 * kGold + stream_mix)));
        float v1 = gamma06(mix64(key ^ mix64((e0 + 1) * kGold + stream_mix)));
        float inv = 1.0f / fmaxf(sqrtf(wave_sum(v0 * v0 + v1 * v1)), 1e-12f);
        v0 = fminf(v0 * inv, 0.2f);
        v1 = fminf(v1 * inv, 0.2f);
        inv = 1.0f / fmaxf(sqrtf(wave_sum(v0 * v0 + v1 * v1)), 1e-12f);
        v0 = fminf(fmaxf(rintf(v0 * inv * 512.0f), 0.0f), 255.0f);
        v1 = fminf(fmaxf(rintf(v1 * inv * 512.0f), 0.0f), 255.0f);
        *(float2*)(out + r * 128 + 2 * lane) = make_float2(v0, v1);
    }
}

__device__ __forceinline__ float normal24(unsigned long long b)
{
    const float u1 = ((float)(b >> 40) + 0.5f) * 0x1p-24f, u2 = ((float)(b & 0xFFFFFF) + 0.5f) * 0x1p-24f;
    return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}

__global__ __launch_bounds__(256)
void noisy_copies_kernel(const long long* __restrict__ keys, const float* __restrict__ query, int nq, const long long* __restrict__ src,
                         const long long* __restrict__ dst, int nmatch, long rows, unsigned long long stream_mix, float sigma,
                         float* __restrict__ train, int nt)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nwaves) {
        const long b = r / nmatch, m = r - b * nmatch;
        const long s = src[r], d = dst[r];
        if (s < 0 || s >= nq || d < 0 || d >= nt) continue;                 // wave-uniform
        const unsigned long long key = (unsigned long long)keys[b];
        const unsigned long long e0 = (unsigned long long)(m * 128 + 2 * lane);
        const float2 q = *(const float2*)(query + ((size_t)b * nq + s) * 128 + 2 * lane);
        const float v0 = q.x + sigma * normal24(mix64(key ^ mix64(e0 * kGold + stream_mix)));
        const float v1 = q.y + sigma * normal24(mix64(key ^ mix64((e0 + 1) * kGold + stream_mix)));
        *(float2*)(train + ((size_t)b * nt + d) * 128 + 2 * lane) =
            make_float2(fminf(fmaxf(rintf(v0), 0.0f), 255.0f), fminf(fmaxf(rintf(v1), 0.0f), 255.0f));
    }
}

// the hash words themselves: out[b, i] = mix(key[b] ^ mix(i * GOLD + mix(stream))), what PairRng.bits computes with a
// dozen elementwise tensor ops -- the small per-pair draws of the scene generator are launch bound without it
__global__ __launch_bounds__(256)
void hash64_kernel(const long long* __restrict__ keys, long n, long total, unsigned long long stream_mix, long long* __restrict__ out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / n, e = i - b * n;
        out[i] = (long long)mix64((unsigned long long)keys[b] ^ mix64((unsigned long long)e * kGold + stream_mix));
    }
}

}  // namespace

extern "C" int vo_synth_hash64(const int64_t* keys, int batch, long n, uint64_t stream_id, int64_t* out, void* stream)
{
    VO_CHECK_ARG(batch >= 0 && n >= 0, VO_EINVAL, "synth: batch=%d n=%ld", batch, n);
    const long total = (long)batch * n;
    if (total == 0) return VO_OK;
    VO_CHECK_ARG(keys && out, VO_EINVAL, "synth: null pointer");
    const int grid = (int)std::min<long>((total + 255) / 256, 256L * 16);
    hipLaunchKernelGGL(hash64_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)keys, n, total,
                       mix64(stream_id), (long long*)out);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_synth_sift_like(const int64_t* keys, int batch, int n, uint64_t stream_id, float* out, void* stream)
{
    VO_CHECK_ARG(batch >= 0 && n >= 0, VO_EINVAL, "synth: batch=%d n=%d", batch, n);
    const long rows = (long)batch * n;
    if (rows == 0) return VO_OK;
    VO_CHECK_ARG(keys && out, VO_EINVAL, "synth: null pointer");
    const int grid = (int)std::min<long>((rows + 3) / 4, 256L * 32);
    hipLaunchKernelGGL(sift_like_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)keys, n, rows,
                       mix64(stream_id), out);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_synth_noisy_copies(const int64_t* keys, int batch, const float* query, int nq, const int64_t* src,
                                     const int64_t* dst, int nmatch, uint64_t stream_id, float sigma, float* train, int nt,
                                     void* stream)
{
    VO_CHECK_ARG(batch >= 0 && nq >= 0 && nt >= 0 && nmatch >= 0, VO_EINVAL, "synth: batch=%d nq=%d nt=%d nmatch=%d", batch, nq, nt, nmatch);
    const long rows = (long)batch * nmatch;
    if (rows == 0) return VO_OK;
    VO_CHECK_ARG(keys && query && src && dst && train, VO_EINVAL, "synth: null pointer");
    const int grid = (int)std::min<long>((rows + 3) / 4, 256L * 32);
    hipLaunchKernelGGL(noisy_copies_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)keys, query, nq,
                       (const long long*)src, (const long long*)dst, nmatch, rows, mix64(stream_id), sigma, train, nt);
    VO_LAUNCH_CHECK();
    return VO_OK;
}
