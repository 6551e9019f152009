// Measurement aid (not product code): numerics of v_mfma_f32_32x32x16_f16 on gfx950, as the F scoring kernel's
// f16-split filter relies on them.  Questions: (1) are f16 subnormal A/B inputs honoured (not flushed)?
// (2) how far is a K = 32 chain (two MFMAs) from the exactly rounded dot product, in units of 2^-24 * sum|terms|,
// on operands with heavy cancellation?  (3) which simple model reproduces the bits: a k-ordered fmaf chain, exact
// groups of 4 products added to the accumulator, or one rounding per MFMA?
// Build: hipcc --offload-arch=gfx950 -O2 -w scripts/ubench/mfma_f16_numerics.hip -o scripts/ubench/mfma_f16_numerics
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A: 32 x K row-major, B: K x 32 stored as Bt 32 x K (column j's K entries contiguous), D: 32 x 32
template <int KSTEPS>
__global__ __launch_bounds__(64) void k(const _Float16* A, const _Float16* Bt, float* D)
{
    const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    v16f acc = {0};
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const v8h a = *(const v8h*)(A + i * 16 * KSTEPS + 16 * s + 8 * kh);
        const v8h b = *(const v8h*)(Bt + i * 16 * KSTEPS + 16 * s + 8 * kh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * kh + (r % 4)) * 32 + i] = acc[r];
}

static double rnd() { return (double)rand() / RAND_MAX; }

int main()
{
    constexpr int KS = 2, K = 16 * KS;
    std::vector<_Float16> A(32 * K), Bt(32 * K);
    std::vector<float> D(1024);
    _Float16 *dA, *dB;
    float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, Bt.size() * 2); hipMalloc(&dD, 4096);
    // (1) subnormals: A = 2^-20 (f16 subnormal), B = 2^10  ->  product 2^-10 each, sum over K
    for (auto& v : A) v = (_Float16)ldexp(1., -20);
    for (auto& v : Bt) v = (_Float16)1024.;
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
    k<KS><<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    printf("subnormal A: D[0] = %g (expected %g if honoured, 0 if flushed)\n", D[0], K * ldexp(1., -10));
    for (auto& v : A) v = (_Float16)1024.;
    for (auto& v : Bt) v = (_Float16)ldexp(1., -24);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
    k<KS><<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    printf("subnormal B: D[0] = %g (expected %g if honoured, 0 if flushed)\n", D[0], K * ldexp(1., -14));

    // (2,3) cancellation-heavy random operands
    double worst = 0, worst_rel_exact = 0;
    long match_chain = 0, match_g4 = 0, match_g8 = 0, match_one = 0, match_exact = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        srand(1234 + trial);
        for (int i = 0; i < 32; ++i)
            for (int kk = 0; kk < K; ++kk) {
                const int e = trial % 3 == 0 ? 0 : (rand() % 12) - 6;
                A[i * K + kk] = (_Float16)((rnd() * 2 - 1) * ldexp(1., e));
                Bt[i * K + kk] = (_Float16)((rnd() * 2 - 1) * ldexp(1., (trial % 3 == 1) ? (rand() % 10) : 0));
            }
        if (trial % 2) {   // force cancellation: make pairs of terms nearly opposite
            for (int i = 0; i < 32; ++i)
                for (int kk = 0; kk + 1 < K; kk += 2) { A[i * K + kk + 1] = -A[i * K + kk]; }
            for (int j = 0; j < 32; ++j)
                for (int kk = 0; kk + 1 < K; kk += 2) { Bt[j * K + kk + 1] = (_Float16)((double)Bt[j * K + kk] * (1 + ldexp(1., -9))); }
        }
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
        k<KS><<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ex = 0, mag = 0;
                float chain = 0.f, g4 = 0.f, g8 = 0.f, one = 0.f;
                for (int s = 0; s < KS; ++s) {
                    double dot16 = 0;
                    for (int g = 0; g < 4; ++g) {
                        double dot4 = 0;
                        for (int q = 0; q < 4; ++q) {
                            const int kk = 16 * s + 4 * g + q;
                            const double p = (double)A[i * K + kk] * (double)Bt[j * K + kk];
                            ex += p; mag += fabs(p); dot4 += p;
                            chain = fmaf((float)A[i * K + kk], (float)Bt[j * K + kk], chain);
                        }
                        g4 = (float)((double)g4 + dot4);
                        dot16 += dot4;
                        if (g & 1) { /* groups of 8 */ }
                    }
                    double d8a = 0, d8b = 0;
                    for (int q = 0; q < 8; ++q) {
                        d8a += (double)A[i * K + 16 * s + q] * (double)Bt[j * K + 16 * s + q];
                        d8b += (double)A[i * K + 16 * s + 8 + q] * (double)Bt[j * K + 16 * s + 8 + q];
                    }
                    g8 = (float)((double)g8 + d8a);
                    g8 = (float)((double)g8 + d8b);
                    one = (float)((double)one + dot16);
                }
                const float got = D[i * 32 + j];
                total++;
                match_chain += got == chain; match_g4 += got == g4; match_g8 += got == g8; match_one += got == one;
                match_exact += got == (float)ex;
                if (mag > 0) {
                    worst = fmax(worst, fabs((double)got - ex) / (ldexp(1., -24) * mag));
                    if (ex != 0) worst_rel_exact = fmax(worst_rel_exact, fabs((double)got - ex) / fabs(ex));
                }
            }
    }
    printf("K = %d chain of %d MFMAs, %ld outputs\n", K, KS, total);
    printf("bitwise matches: fmaf chain %ld, exact groups of 4 %ld, groups of 8 %ld, one rounding per MFMA %ld, exactly rounded %ld\n",
           match_chain, match_g4, match_g8, match_one, match_exact);
    printf("worst |D - exact| = %.3f x 2^-24 x sum|terms|   (worst relative to |exact|: %.3g)\n", worst, worst_rel_exact);
    return 0;
}
