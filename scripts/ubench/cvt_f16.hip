// Measurement aid: do v_cvt_f16_f32 and v_fma_mixlo_f16 produce f16 subnormals on gfx950 under hipcc's default mode?
// Build: hipcc --offload-arch=gfx950 -O2 -w scripts/ubench/cvt_f16.hip -o scripts/ubench/cvt_f16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const float* x, uint32_t* out, int n)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    uint32_t a, b;
    const float v = x[i], one = 1.f;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(v));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(b) : "v"(v), "v"(one), "0"(0u));
    out[2 * i] = a & 0xffff;
    out[2 * i + 1] = b & 0xffff;
}
int main()
{
    float h[8] = {ldexpf(1.f, -14), ldexpf(1.f, -15), ldexpf(1.f, -20), ldexpf(3.f, -24), ldexpf(1.f, -24), ldexpf(1.f, -25) * 1.5f, 0.1f, 1e-5f};
    float* d; uint32_t* o; uint32_t r[16];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, 8);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) {
        _Float16 e = (_Float16)h[i];
        uint16_t eb; __builtin_memcpy(&eb, &e, 2);
        printf("x = %-12g host f16 bits %04x   v_cvt_f16_f32 %04x   v_fma_mixlo_f16 %04x\n", h[i], eb, r[2 * i], r[2 * i + 1]);
    }
    return 0;
}
