// In-kernel shader clock and issue cycles per wave64 VALU instruction on gfx950 (not product code).
// Reconciles "2.8-2.9 cycles per v_fma_f32 at the nominal 2.4 GHz" (wall-clock ubench) with
// the 2 cycles of MI355X_MICROARCH.md: every wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) around its instruction loop, so clock = d(memtime)/d(memrealtime) x 100 MHz and
// cycles per instruction per SIMD = d(memtime) / (instructions the SIMD's resident waves issued).
//   hipcc --offload-arch=gfx950 -O3 -o clock scripts/ubench/clock.hip && ./clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters, float seed)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
    const float m = 1.0000001f, c = 1e-9f;
    const int si = (int)seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"(m));
                if (OP == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"(m));
                if (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(*(float*)&a[i]), "v"(m) : "vcc");
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(float*)&a[i]) : "v"(m));
                if (OP == 5) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(*(int*)&a[i]) : : "vcc");
                if (OP == 6) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(si));
                if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 9, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 8) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(si));
                if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 10) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"((double)m), "v"((double)c));
                if (OP == 11) asm volatile("v_max_i32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 12) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(si));
                if (OP == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"((double)m), "v"((double)c));
                if (OP == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 15) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(*(int*)&a[i]));
                if (OP == 16) asm volatile("v_add_co_u32 %0, vcc, %0, %0" : "+v"(*(int*)&a[i]) : : "vcc");
                if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 31, 1" : "+v"(*(int*)&a[i]));
                if (OP == 18) asm volatile("v_mul_f32 %0, %0, %1 clamp" : "+v"(*(float*)&a[i]) : "v"(m));
                if (OP == 19) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(*(float*)&a[i]), "v"(m) : "s20", "s21");
                if (OP == 20) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"((double)m));
                if (OP == 21) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"((double)c));
                if (OP == 22) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                if (OP == 23) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(*(float*)&a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 24) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(m));
                if (OP == 25) asm volatile("v_rcp_f32 %0, %0" : "+v"(*(float*)&a[i]));
                if (OP == 26) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(it), "v"(si));
                if (OP == 27) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 28) asm volatile("v_fma_f32 %0, %0, %1, |%2|" : "+v"(*(float*)&a[i]) : "v"(m), "v"(c));
                if (OP == 29) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[i]) : "s"(m), "v"(c));
                if (OP == 30) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(*(float*)&a[i]) : "v"(m), "v"(c));
                if (OP == 31) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(*(float*)&a[i]) : "v"(c));
                if (OP == 32) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"((double)m), "v"((double)c));
                if (OP == 33) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %1" : "+v"(a[i]) : "v"((double)m) : "vcc");
                if (OP == 34) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"((double)m), "v"((double)c));
                if (OP == 35) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(it));
                if (OP == 36) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(si), "v"(it));
                if (OP == 37) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(si));
                if (OP == 38) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(*(int*)&a[i]) : "v"(si), "v"(it));
                if (OP == 39) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(*(float*)&a[i]) : "v"(si), "v"(it));
                if (OP == 40) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(*(int*)&a[i]) : "v"(si));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char* name, int wg_per_cu)
{
    const int cus = 256, iters = 20000, wgs = cus * wg_per_cu;
    float* out;
    unsigned long long* st;
    hipMalloc(&out, sizeof(float) * 256 * wgs);
    hipMalloc(&st, sizeof(unsigned long long) * 8 * wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 30; ++rep) hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, out, st, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, out, st, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8 * wgs);
    hipMemcpy(h.data(), st, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    const double instr_per_wave = (double)iters * 64;
    for (int w = 0; w < 4 * wgs; ++w) {
        clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);          // GHz
        cyc.push_back((double)h[2 * w] / (instr_per_wave * wg_per_cu));         // per SIMD: wg_per_cu waves share it
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double wall_ns = ms * 1e6 / (instr_per_wave * wg_per_cu);
    printf("%-16s waves/SIMD %d: wall %.3f ns/instr/SIMD (= %.2f cyc @2.4GHz nominal); in-kernel clock %.3f GHz (median), "
           "%.2f shader cycles/instr/SIMD (median)\n", name, wg_per_cu, wall_ns, wall_ns * 2.4, clk[clk.size() / 2],
           cyc[cyc.size() / 2]);
    hipFree(out); hipFree(st);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
    }
    const int w = 4;
    run<1>("v_mul_f32", w); run<2>("v_min_f32", w); run<3>("v_cmp_lt_f32", w); run<4>("v_cndmask_b32", w);
    run<5>("v_addc_co_u32", w); run<6>("v_max3_i32", w); run<7>("v_lshl_add_u32", w); run<8>("v_med3_i32", w);
    run<9>("v_add_u32", w); run<10>("v_fma_f64", w); run<11>("v_max_i32", w); run<12>("v_or3_b32", w); run<13>("v_pk_fma_f32", w);
    run<14>("v_alignbit_b32", w); run<15>("v_lshrrev_b32", w); run<16>("v_add_co_u32", w); run<17>("v_bfe_u32", w);
    run<18>("v_mul_f32 clamp", w); run<19>("v_cmp_lt_f32 sgpr", w); run<20>("v_mul_f64", w); run<21>("v_add_f64", w);
    run<22>("v_rcp_f64", w); run<23>("v_cvt_f32_f64", w); run<24>("v_cvt_f64_f32", w); run<25>("v_rcp_f32", w);
    run<26>("v_and_or_b32", w); run<27>("v_lshl_or_b32", w); run<28>("v_fma_f32 |abs|", w); run<29>("v_fma_f32 sgpr", w);
    run<30>("v_fmac_f32", w); run<31>("v_sub_f32", w); run<32>("v_div_fmas_f64", w); run<33>("v_div_scale_f64", w);
    run<34>("v_div_fixup_f64", w); run<35>("v_bcnt_u32_b32", w);
    run<36>("v_pk_fma_f16", w); run<37>("v_pk_mul_f16", w); run<38>("v_fma_f16", w); run<39>("v_dot2_f32_f16", w); run<40>("v_pk_max_f16", w);
    return 0;
}
