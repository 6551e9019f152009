// Micro-benchmark (measurement aid, not product code): wall-clock int8 MFMA throughput of the whole chip under its power
// management, v_mfma_i32_32x32x32_i8 against v_mfma_i32_16x16x64_i8 (same multiply-adds per "unit": 4 of the former, 8 of
// the latter), bare and with the kNN epilogue's 36 integer VALU instructions per unit interleaved.  DESIGN.md 4.1 left the
// 16x16x64 form untried; MI355X_MICROARCH.md reports the 16x16 bf16 shape clocking higher under DVFS.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_shape_power.hip -o scripts/ubench/mfma_shape_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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

// MODE 0: 32x32x32 bare, 1: 16x16x64 bare, 2: 32x32x32 + VALU, 3: 16x16x64 + VALU
template <int MODE>
__global__ __launch_bounds__(512) void k(const v4i* __restrict__ data, int* out, int iters)
{
    // four A and four B fragments per lane (distinct values: the multiplier inputs toggle as in a real sweep)
    v4i a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = data[(threadIdx.x + 512 * j) & 4095];
        b[j] = data[(threadIdx.x * 7 + 64 * j + 1) & 4095];
    }
    v16i big[2] = {{0}, {0}};
    v4i small[4] = {{0}, {0}, {0}, {0}};
    int m1 = threadIdx.x, m2 = 1, c = 2, x = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                big[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], b[j], big[0], 0, 0, 0);
                if (MODE == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    VALU9(m1, m2, x, c)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                small[j & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j & 3], b[(j + (j >> 2)) & 3], small[j & 3], 0, 0, 0);
                if (MODE == 3 && (j & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    VALU9(m1, m2, x, c)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    int s = m1 ^ m2 ^ c;
#pragma unroll
    for (int r = 0; r < 16; ++r) s ^= big[0][r] ^ big[1][r];
#pragma unroll
    for (int j = 0; j < 4; ++j) s ^= small[j][0] ^ small[j][1] ^ small[j][2] ^ small[j][3];
    if (s == 0x12345678) out[0] = s;
}

template <int MODE>
static void run(const char* name, const v4i* data, int* out, int waves_per_simd)
{
    const int iters = 100000, grid = 256 * waves_per_simd / 2;     // 512 threads = 8 waves = 2 per SIMD per workgroup
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, data, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 2.0 * 4 * 32768.0 * iters * (double)grid * 8;     // per unit: 4 x 32x32x32 multiply-adds, 2 ops each
        if (rep == 2) printf("%-28s %d waves/SIMD: %8.2f ms  %6.3f POP/s  (%.3f of 5.033)\n", name, waves_per_simd, ms, ops / ms / 1e12, ops / ms / 1e12 / 5.033);
    }
}

int main()
{
    std::vector<int> h(4096 * 4);
    srand(7);
    // SIFT-like bytes: small magnitudes offset by -128, as the product encodes them
    for (auto& w : h) {
        unsigned v = 0;
        for (int q = 0; q < 4; ++q) {
            const int u = (rand() % 100 < 7) ? 0 : (rand() % 64) * (rand() % 3 == 0 ? 2 : 1);
            v |= (unsigned)((u - 128) & 0xff) << (8 * q);
        }
        w = (int)v;
    }
    v4i* d; int* out;
    hipMalloc(&d, h.size() * 4); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wps : {2, 4}) {
        run<0>("32x32x32 bare", d, out, wps);
        run<1>("16x16x64 bare", d, out, wps);
        run<2>("32x32x32 + 36 VALU/unit", d, out, wps);
        run<3>("16x16x64 + 36 VALU/unit", d, out, wps);
    }
    return 0;
}
