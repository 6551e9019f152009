// f64 VALU issue-rate microbenchmark for gfx950 (not product code): cycles per wave64 instruction per
// SIMD for v_fma_f64 / v_mul_f64 / v_add_f64 / v_cmp_f64 at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o f64rates scripts/ubench/f64rates.hip && ./f64rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double m = 1.0000001, c = 1e-9;
    const int seed_i = (int)seed;
    int cnt = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"((float)m), "v"((float)c));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(*(float*)&a[i]), "v"((float)m) : "vcc");
                if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"((float)m));
                if (OP == 9) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(seed_i));
                if (OP == 10) asm volatile("v_max_i32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 11) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"((float)m), "v"((float)c));
                if (OP == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"((float)m));
                if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 9, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 14) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(*(int*)&a[i]));
                if (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 16) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(seed_i));
                if (OP == 17) asm volatile("v_lshlrev_b32 %0, 9, %0" : "+v"(*(int*)&a[i]));
                if (OP == 18) asm volatile("v_or_b32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 19) asm volatile("v_max_u32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 20) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(seed_i));
                if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(seed_i));
                if (OP == 22) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 23) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"((float)m));
                if (OP == 25) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
            }
        }
    }
    double s = cnt;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int wg_per_cu)
{
    int cus = 256;
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * cus * wg_per_cu);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * wg_per_cu;   // 4 waves per WG = 1 per SIMD
    printf("%-12s waves/SIMD %d: %.2f ns per instr per SIMD = %.2f cycles @2.4GHz\n", name, wg_per_cu,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {4}) {
        run<0>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_cmp_lt_f64", w); run<4>("v_fma_f32", w); run<5>("v_pk_fma_f32", w); run<6>("v_pk_mul_f32", w); run<7>("v_cmp_lt_f32", w); run<8>("v_mul_f32", w);
        run<9>("v_med3_i32", w); run<10>("v_max_i32", w); run<11>("v_med3_f32", w); run<12>("v_max_f32", w); run<13>("v_lshl_add_u32", w); run<14>("v_cvt_f32_i32", w); run<15>("v_add_u32", w);
        run<16>("v_mad_i32_i24", w); run<17>("v_lshlrev_b32", w); run<18>("v_or_b32", w); run<19>("v_max_u32", w); run<20>("v_max3_i32", w);
        run<21>("v_add3_u32", w); run<22>("v_sub_u32", w); run<23>("v_cndmask_b32", w); run<25>("v_pk_max_i16", w);
    }
    return 0;
}
