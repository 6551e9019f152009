// Micro-benchmark (measurement aid, not product code): issue rates of the instructions the kNN
// sweep is made of, per SIMD, at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed, 7, 9};
    v16i acc0 = {0}, acc1 = {0};
    int m1 = seed, m2 = seed + 1, m3 = seed + 2, m4 = seed + 3, x = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // 8 MFMA, two independent chains
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
            }
        } else if (MODE == 1) {     // 8 MFMA, one dependent chain
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        } else if (MODE == 2) {     // 96 VALU: lshl_add, med3, max (two independent chains)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int p, q;
                asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p) : "v"(x), "v"(m3));
                asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m1), "v"(m2), "v"(p));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(p));
                asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(q) : "v"(x), "v"(m1));
                asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m4) : "v"(m3), "v"(m4), "v"(q));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(m3) : "v"(m3), "v"(q));
            }
        } else if (MODE == 3) {     // 4 MFMA (dependent) + 48 VALU interleaved 1:12
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int p;
                    asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p) : "v"(x), "v"(m3));
                    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m1), "v"(m2), "v"(p));
                    asm volatile("v_max_i32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(p));
                }
            }
        } else if (MODE == 4) {     // 4 MFMA (dependent) then 48 VALU on the result
#pragma unroll
            for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int p;
                asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(p) : "v"(acc0[r]), "v"(m3));
                asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(m2) : "v"(m1), "v"(m2), "v"(p));
                asm volatile("v_max_i32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(p));
            }
            acc0 = (v16i){0};
        }
    }
    int s = m1 + m2 + m3 + m4;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 0x12345678) out[0] = s;
}

template <int MODE>
void run(const char* name, int wg_per_cu, int per_iter_mfma, int per_iter_valu)
{
    int* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 100, 1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 1);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = wg_per_cu;       // 256 threads = 4 waves = 1 per SIMD
    const double ns_per_iter_simd = ms * 1e6 / iters;   // all SIMDs run concurrently
    printf("%-34s waves/SIMD %d: %.1f ns per iteration per SIMD -> per wave-iteration %.1f ns; at 2.4 GHz: %.0f cycles per SIMD-iteration-slot (%d MFMA, %d VALU per wave-iteration)\n",
           name, wg_per_cu, ns_per_iter_simd, ns_per_iter_simd / waves_per_simd, ns_per_iter_simd / waves_per_simd * 2.4, per_iter_mfma, per_iter_valu);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("8 MFMA, 2 chains", w, 8, 0);
        run<1>("8 MFMA, 1 chain", w, 8, 0);
        run<2>("96 VALU (lshl_add/med3/max)", w, 0, 96);
        run<3>("4 MFMA + 48 VALU interleaved", w, 4, 48);
        run<4>("4 MFMA then 48 dependent VALU", w, 4, 48);
    }
    return 0;
}
