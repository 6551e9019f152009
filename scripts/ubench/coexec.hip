// Micro-benchmark (measurement aid, not product code): do v_mfma_i32_32x32x32_i8 and the integer VALU instructions of
// the kNN top-2 epilogue (v_lshl_add_u32 / v_med3_i32 / v_max3_i32) execute side by side on one SIMD of gfx950?
// Every wave stamps s_memtime around its loop; printed: shader cycles per loop iteration per wave (median over waves),
// at 1 to 4 waves per SIMD (one workgroup per CU).  One iteration = 4 dependent MFMAs and/or 36 VALU instructions.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/coexec.hip -o scripts/ubench/coexec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VALU9(m1, m2, x, c)                                                                  \
    {                                                                                        \
        int p0, p1, p2, p3, t0, t1;                                                          \
        asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p0) : "v"(x), "v"(c));            \
        asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p1) : "v"(c), "v"(x));            \
        asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(t0) : "v"(m1), "v"(p0), "v"(p1));    \
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(p0), "v"(p1));    \
        asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p2) : "v"(x), "v"(m1));           \
        asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p3) : "v"(c), "v"(m1));           \
        asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(t1) : "v"(m2), "v"(p2), "v"(p3));    \
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m2), "v"(p2), "v"(p3));    \
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(t0), "v"(t1));      \
    }
#define ADD9(m1, m2, x, c)                                                                   \
    {                                                                                        \
        _Pragma("unroll") for (int q = 0; q < 9; ++q) asm volatile("v_add_u32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(x)); \
    }

template <int MODE>
__global__ __launch_bounds__(1024) void k(long long* stamps, int* out, int iters, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed, 7, 9};
    v16i acc[3] = {{0}, {0}, {0}};
    int m1 = seed, m2 = seed + 1, c = seed + 2, x = threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < (int)(blockDim.x >> 7);          // MODE 3: first half of the waves
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) VALU9(m1, m2, x, c)
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU9(m1, m2, x, c)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 3) {
            if (mfma_wave) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) VALU9(m1, m2, x, c)
            }
        } else if (MODE == 4) {            // the kernel's dataflow: three rotating accumulators, VALU reads the set finished a unit ago
#pragma unroll
            for (int u = 0; u < 3; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v16i z = {0};
                    acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, j == 0 ? z : acc[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const v16i& r = acc[(u + 1) % 3];
                    int p0 = (r[4 * j] << 9) + c, p1 = (r[4 * j + 1] << 9) + x, p2 = (r[4 * j + 2] << 9) + c, p3 = (r[4 * j + 3] << 9) + x, t0_, t1_;
                    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(t0_) : "v"(m1), "v"(p0), "v"(p1));
                    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(p0), "v"(p1));
                    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(t1_) : "v"(m2), "v"(p2), "v"(p3));
                    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m2), "v"(p2), "v"(p3));
                    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(t0_), "v"(t1_));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ADD9(m1, m2, x, c)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ADD9(m1, m2, x, c)
        } else if (MODE == 7) {           // 18 VALU per MFMA
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU9(m1, m2, x, c)
                VALU9(m2, m1, c, x)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 8) {           // 4 VALU per MFMA
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                int p0, p1;
                asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p0) : "v"(x), "v"(c));
                asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p1) : "v"(c), "v"(x));
                asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m1), "v"(p0), "v"(p1));
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(p0), "v"(p1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (blockIdx.x * (blockDim.x >> 6) + wave)] = t1 - t0;
        stamps[2 * (blockIdx.x * (blockDim.x >> 6) + wave) + 1] = (MODE == 3 && !mfma_wave) ? 1 : 0;
    }
    int s = m1 + m2 + c;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[u][r];
    if (s == 0x12345678) out[0] = s;
}

template <int MODE>
void run(const char* name, int threads, double iter_scale)
{
    const int blocks = 256, waves = blocks * threads / 64, iters = 20000;
    long long* d; int* o;
    hipMalloc(&d, waves * 16); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, o, 2000, 1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, 1);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int w = 0; w < waves; ++w) (h[2 * w + 1] ? b : a).push_back((double)h[2 * w] / iters / iter_scale);
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-58s %d waves/SIMD: %7.1f cycles per unit per wave", name, threads / 256, a[a.size() / 2]);
    if (!b.empty()) printf(" (MFMA waves), %7.1f (VALU waves)", b[b.size() / 2]);
    printf("; wall %.3f ms -> %.0f MHz\n", ms, a[a.size() / 2] * iter_scale * iters / (ms * 1e3));
    hipFree(d); hipFree(o);
}

int main()
{
    for (int threads : {256, 512, 768, 1024}) {
        run<0>("4 dependent MFMA", threads, 1);
        run<1>("36 VALU (lshl_add/med3/max3)", threads, 1);
        run<2>("4 x (MFMA + 9 VALU), VALU independent of the MFMA", threads, 1);
        run<4>("4 x (MFMA + 9 VALU), VALU reads an older accumulator", threads, 3);
        run<7>("4 x (MFMA + 18 VALU)", threads, 1);
        run<8>("4 x (MFMA + 4 VALU)", threads, 1);
        run<6>("36 v_add_u32", threads, 1);
        run<5>("4 x (MFMA + 9 v_add_u32)", threads, 1);
    }
    run<3>("half the waves 8 MFMA, the other half 72 VALU (per 2 units)", 512, 2);
    return 0;
}
