// What limits the tier-1 loop of score_kernel<F>?  (not product code)  The loop body of fund_filter32_signs
// with knobs: points from LDS (ds_read_b128 broadcast) or held in registers, sign bookkeeping on or off,
// one or two models per lane.  Prints shader cycles per evaluation per SIMD (s_memtime) and the clock.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o score_loop scripts/ubench/score_loop.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Flt { float F[9], tlo, thi, eta2, c; };

__device__ __forceinline__ void eval(const Flt& f, float x1, float y1, float x2, float y2, uint32_t& ib, uint32_t& ob, int signs)
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
    const float lo = __builtin_fmaf(d, d, mm), nhi = __builtin_fmaf(-d, d, mm);
    const float r_in = __builtin_fmaf(-f.tlo, dmin, lo);
    const float r_out = __builtin_fmaf(f.thi, dmin, nhi);
    if (signs == 1) {
        ib = __builtin_amdgcn_alignbit(ib, __float_as_uint(r_in), 31);
        ob = __builtin_amdgcn_alignbit(ob, __float_as_uint(r_out), 31);
    } else if (signs == 2) {                      // shift + add (VOP2 only)
        ib += __float_as_uint(r_in) >> 31;
        ob = (ob << 1) + (__float_as_uint(r_out) >> 31);
    } else {                                      // cheapest possible consumer: keeps the FMAs alive
        ib ^= __float_as_uint(r_in);
        ob ^= __float_as_uint(r_out);
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f sp2(float x) { return (v2f){x, x}; }

// two correspondences per lane at once: every FMA of fund_filter32_signs as one v_pk_fma_f32
__device__ __forceinline__ void eval2(const Flt& f, v2f x1, v2f y1, v2f x2, v2f y2, uint32_t& ib, uint32_t& ob)
{
    const float* F = f.F;
    const v2f a1 = pfma(sp2(F[0]), x1, pfma(sp2(F[1]), y1, sp2(F[2])));
    const v2f b1 = pfma(sp2(F[3]), x1, pfma(sp2(F[4]), y1, sp2(F[5])));
    const v2f c1 = pfma(sp2(F[6]), x1, pfma(sp2(F[7]), y1, sp2(F[8])));
    const v2f den2 = pfma(a1, a1, b1 * b1);
    const v2f d = pfma(x2, a1, pfma(y2, b1, c1));
    const v2f a2 = pfma(sp2(F[0]), x2, pfma(sp2(F[3]), y2, sp2(F[6])));
    const v2f b2 = pfma(sp2(F[1]), x2, pfma(sp2(F[4]), y2, sp2(F[7])));
    const v2f den1 = pfma(a2, a2, b2 * b2);
    const v2f dmin = (v2f){fminf(den1.x, den2.x), fminf(den1.y, den2.y)};
    const v2f ad = (v2f){fabsf(d.x), fabsf(d.y)};
    const v2f mm = pfma(sp2(f.eta2), ad, sp2(f.c));
    const v2f lo = pfma(d, d, mm), nhi = pfma(-d, d, mm);
    const v2f r_in = pfma(sp2(-f.tlo), dmin, lo);
    const v2f r_out = pfma(sp2(f.thi), dmin, nhi);
    ib = __builtin_amdgcn_alignbit(ib, __float_as_uint(r_in.x), 31);
    ib = __builtin_amdgcn_alignbit(ib, __float_as_uint(r_in.y), 31);
    ob = __builtin_amdgcn_alignbit(ob, __float_as_uint(r_out.x), 31);
    ob = __builtin_amdgcn_alignbit(ob, __float_as_uint(r_out.y), 31);
}

__global__ __launch_bounds__(256) void kp(const float4* pts, int np, int reps, uint32_t* out, unsigned long long* stamps)
{
    __shared__ float4 sp[512];                     // pair p: sp[2p] = (x1 x1' y1 y1'), sp[2p+1] = (x2 x2' y2 y2')
    for (int i = threadIdx.x; i < np; i += 256) sp[i] = pts[i];
    Flt f;
#pragma unroll
    for (int q = 0; q < 9; ++q) f.F[q] = 0.01f * (q + 1) + 1e-4f * threadIdx.x;
    f.tlo = 0.0099f; f.thi = 0.0101f; f.eta2 = 1e-4f; f.c = 1e-6f;
    __syncthreads();
    uint32_t ib = 0, ob = 0, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int i0 = 0; i0 < np; i0 += 32) {
#pragma unroll 4
            for (int kk = 0; kk < 16; ++kk) {
                const float4 u = sp[i0 + 2 * kk], v = sp[i0 + 2 * kk + 1];
                eval2(f, (v2f){u.x, u.y}, (v2f){u.z, u.w}, (v2f){v.x, v.y}, (v2f){v.z, v.w}, ib, ob);
            }
            acc += __popc(ib) + ob;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * w] = t1 - t0; stamps[4 * w + 1] = q1 - q0; stamps[4 * w + 2] = q0; stamps[4 * w + 3] = q1;
    }
}

template <int LDS, int SIGNS, int MODELS>
__global__ __launch_bounds__(256) void k(const float4* pts, int np, int reps, uint32_t* out, unsigned long long* stamps)
{
    __shared__ float4 sp[512];
    for (int i = threadIdx.x; i < np; i += 256) sp[i] = pts[i];
    Flt f[MODELS];
#pragma unroll
    for (int m = 0; m < MODELS; ++m) {
#pragma unroll
        for (int q = 0; q < 9; ++q) f[m].F[q] = 0.01f * (q + 1) + 1e-4f * threadIdx.x + 0.3f * m;
        f[m].tlo = 0.0099f; f[m].thi = 0.0101f; f[m].eta2 = 1e-4f; f[m].c = 1e-6f;
    }
    __syncthreads();
    uint32_t ib[MODELS] = {}, ob[MODELS] = {};
    uint32_t acc = 0;
    float4 r0 = sp[0], r1 = sp[1], r2 = sp[2], r3 = sp[3];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int i0 = 0; i0 < np; i0 += 32) {
#pragma unroll 4
            for (int kk = 0; kk < 32; ++kk) {
                float4 v;
                if (LDS) v = sp[i0 + kk];
                else { v = (kk & 3) == 0 ? r0 : (kk & 3) == 1 ? r1 : (kk & 3) == 2 ? r2 : r3; }
#pragma unroll
                for (int m = 0; m < MODELS; ++m) eval(f[m], v.x, v.y, v.z, v.w, ib[m], ob[m], SIGNS);
            }
#pragma unroll
            for (int m = 0; m < MODELS; ++m) { acc += __popc(ib[m]) + ob[m]; }
            if (!LDS) { r0.x += 1e-3f; r1.y += 1e-3f; r2.z += 1e-3f; r3.w += 1e-3f; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * w] = t1 - t0;
        stamps[4 * w + 1] = q1 - q0;
        stamps[4 * w + 2] = q0;
        stamps[4 * w + 3] = q1;
    }
}

typedef void (*kern_t)(const float4*, int, int, uint32_t*, unsigned long long*);
void run_k(kern_t kern, int MODELS, const char* name, int wg_per_cu)
{
    const int cus = 256, wgs = cus * wg_per_cu, np = 512, reps = 40;
    std::vector<float4> h(np);
    for (int i = 0; i < np; ++i) h[i] = make_float4(100.f + i, 50.f + 0.5f * i, 101.f + i, 49.f + 0.5f * i);
    float4* pts; uint32_t* out; unsigned long long* st;
    hipMalloc(&pts, np * sizeof(float4)); hipMalloc(&out, 4 * 256 * wgs); hipMalloc(&st, 128 * wgs);
    hipMemcpy(pts, h.data(), np * sizeof(float4), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, pts, np, reps, out, st);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, pts, np, reps, out, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(16 * wgs);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    const double evals_per_wave = (double)np * reps * MODELS;
    unsigned long long first = ~0ull, last = 0, last_start = 0;
    for (int w = 0; w < 4 * wgs; ++w) {
        clk.push_back((double)s[4 * w] / (double)s[4 * w + 1] * 0.1);
        cyc.push_back((double)s[4 * w] / (evals_per_wave * wg_per_cu));
        first = std::min(first, s[4 * w + 2]); last = std::max(last, s[4 * w + 3]); last_start = std::max(last_start, s[4 * w + 2]);
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const size_t n = cyc.size();
    printf("%-44s waves/SIMD %d: cycles/eval/SIMD p5 %.1f p50 %.1f p95 %.1f max %.1f, clock %.3f GHz, event wall %.2f ns/eval/SIMD; "
           "first start -> last end %.3f ms, last start at +%.3f ms, event %.3f ms\n",
           name, wg_per_cu, cyc[n / 20], cyc[n / 2], cyc[n * 19 / 20], cyc[n - 1], clk[n / 2], ms * 1e6 / (evals_per_wave * wg_per_cu),
           (last - first) * 1e-5, (last_start - first) * 1e-5, ms);
    hipFree(pts); hipFree(out); hipFree(st);
}

int main()
{
    for (int w : {2, 4, 7}) {
        run_k(k<1, 1, 1>, 1, "LDS points, alignbit signs, 1 model/lane", w);
        run_k(kp, 1, "packed: 2 correspondences per v_pk_fma_f32", w);
    }
    return 0;
}
