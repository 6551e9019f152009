// Micro-benchmark (measurement aid, not product code): does v_mfma_f32_32x32x2_f32 run beside f32 VALU
// work of the same SIMD?  Modes: MFMA only, VALU only, both interleaved.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a = seed, b = seed + 1.f;
    v16f acc0 = {0}, acc1 = {0};
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < (NM > 0 ? NM : 1); ++j) {
            if (NM > 0) {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NV / (NM > 0 ? NM : 1) / 4; ++t) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    float s = x0 + x1 + x2 + x3;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[0] = s;
}

template <int NM, int NV>
void run(int wg_per_cu)
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((k<NM, NV>), dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 50, 1.f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 1.f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e6 / iters * 2.4;      // per SIMD, all waves of the SIMD together
    printf("%2d MFMA + %3d v_fma_f32 per wave-iteration, %d waves/SIMD: %.0f cycles per SIMD per round of wave-iterations = %.0f per wave-iteration\n",
           NM, NV, wg_per_cu, cyc, cyc / wg_per_cu);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<13, 0>(1); run<0, 256>(1); run<13, 208>(1); run<13, 416>(1); }
        if (w == 2) { run<13, 0>(2); run<0, 256>(2); run<13, 208>(2); run<13, 416>(2); }
        if (w == 4) { run<13, 0>(4); run<0, 256>(4); run<13, 208>(4); run<13, 416>(4); }
    }
    return 0;
}
