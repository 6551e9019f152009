// vo_score_fmfma.h -- the F count kernel with its first tier on the matrix cores.  Included by vo_score.hip INSIDE its
// anonymous namespace, after the scalar filters (it reuses fund_filter32, fund_decide64, ScoreArgs, lds_load).
//
// What moves to MFMA.  The one-sided sure-outlier test of score_kernel<kFund> (fund_filter32_outlier_sign) needs, per
// (model, correspondence), d = x2^T F x1 and den2 = a^2 + b^2 with (a b .) = F x1.  Both are inner products of a
// per-MODEL coefficient vector with a per-POINT feature vector:
//   d    = sum_k c_k phi_k,   phi = (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1),  c = F row-major
//   den2 = sum_k G_k psi_k,   psi = (x1^2, x1 y1, y1^2, x1, y1, 1),  G = the quadratic form of rows 0,1 of F
// so a 32 x 32 tile of evaluations is a small GEMM.  f32 MFMA runs at the VALU's rate and would gain nothing; f16 MFMA
// is 16x faster but f16 has 11 bits, so every factor is split into hi + lo halves (v ~ hi + lo, |v - hi - lo| <=
// 2^-22 |v|) and the products hi*hi + hi*lo + lo*hi are laid along K (the MFMA multiplies f16 pairs exactly and sums in
// f32).  d takes 8 x 3 + 2 = 26 K entries (two K=16 steps; the constant term c_8 = hi + lo meets a (1, 1, 0...) vector
// in the second step's upper lane half), den2 takes 5 x 3 + 1 = 16 (one step): three v_mfma_f32_32x32x16_f16 per 1024
// evaluations, 0.094 cycles per evaluation per SIMD against 0.44 for the 14 VALU instructions of the scalar form at its
// issue peak.  The VALU keeps two instructions per evaluation: r = s - d^2 as one FMA, its sign bit shifted into a bit
// string (v_alignbit_b32); sign set <=> sure outlier.
//
// Error budget of d (unit u = 2^-24 S~, S~ = sum_k |c~_k| B_k + |c~_8| with B_k the problem-wide bound of |phi~_k|):
// f32 rounding of the product features 1; the splits of features and coefficients and the dropped lo*lo product, 2^-22
// relative each: 12; f16 subnormal quantum of the lo halves (2^-25 absolute per factor): <= 2 (+ 2^-22 absolute);
// split of c~_8: 3; accumulation inside the MFMAs: measured <= 3.2 on cancellation-heavy operands
// (scripts/ubench/mfma_f16_numerics.hip, which also shows f16 subnormal operands are honoured), budgeted 30.  Total 48 u
// (+ 2^-20 absolute) = eta.  den2 needs far less: it only has to be an upper bound good to ~2^-10 relative (it moves the
// threshold, not the residual); beta = 2^-17 T~ + 2^-20 (T~ the same kind of bound for the quadratic form; measured
// worst 14 x 2^-24 T~) is added to its constant term, which is then rounded UP to f16.  With thi = thr2 (1 + 2^-16) and
// a per-model dl in [2^-6, 1/2] (2 eta |d| <= dl d^2 + eta^2 / dl):
//   s = (thi den2~ + beta + eta^2 (1/dl - 1)) / (1 - dl)  <  d~^2   =>   (|d| - eta)^2 > thr2 (1 + 2^-17) den2   =>   sure outlier
// (s comes out of the MFMA with all of that folded into its coefficients; the one f32 rounding of r = s - d~^2 moves it by
// < 2^-23 relative, inside the 2^-16).  Everything else -- the survivors' complete f32 filter (tier 1b), the f64 filter
// and the exact evaluation (tier 2) -- is score_kernel<kFund>'s, verbatim.  On the bench data ~2.4 % of the evaluations
// survive, nearly all of them true inliers of their hypothesis.
//
// Layout.  A workgroup = 4 waves = 256 models; a wave owns 64 models = 2 row blocks of 32 (MFMA A operands, 24 VGPRs) for
// the whole kernel and walks chunks of 256 correspondences whose split features sit in LDS as four planes of 16 B per
// point (B operands: lanes 0-31 read plane 0/2, lanes 32-63 plane 1/3 of their column: conflict-free ds_read_b128).
// Models along the rows, correspondences along the columns: a lane's 16 accumulators are 16 models for ONE
// correspondence, so the survivors of a good model are spread over the lanes and the per-lane loops that queue them stay
// short.  128 VGPRs and 40 KB of LDS: 4 workgroups per CU, 4 waves per SIMD -- the queue work is latency-bound and
// needs the other waves to fill its gaps.

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#ifdef VO_SCORE_MF_PROF                  // wave-time per phase (s_memtime ticks) into g_stats[0..5]: scripts/diag/mf_phases.py
#define MF_T(var) const long long var = __builtin_amdgcn_s_memtime()
#define MF_ACC(slot, t0, t1) prof[slot] += (t1) - (t0)
#else
#define MF_T(var)
#define MF_ACC(slot, t0, t1)
#endif
#ifndef VO_SCORE_MF_CHUNK
#define VO_SCORE_MF_CHUNK 256
#endif
#ifndef VO_SCORE_MF_WAVES
#define VO_SCORE_MF_WAVES 4             // waves per SIMD the register budget must allow
#endif
constexpr int kMfChunk = VO_SCORE_MF_CHUNK;
constexpr int kMfEtaUnits = 48;

struct MfModel {
#ifdef VO_SCORE_MF_CHECK
    double dbg_scale, dbg_unit;    // d~ = dbg_scale * d; one budget unit 2^-24 S~ (0 = model outside the tier's domain)
#endif
    v8h ch, cl, c8, s0, s1;    // model-side operands: d coefficients hi / lo, (c~_8 hi, lo, 0...) against the point-side vector (1, 1, 0...);
                               // den2 entries for lane halves 0 / 1
};

// v ~ hi + lo.  The hi half goes through an opaque register: the compiler otherwise converts it twice, by two different
// instruction sequences (v_cvt_pk_f16_f32 of the rounded f32 product for the stored half, v_fma_mixlo_f16 of the factors
// for the one the residual is taken against), which disagree at near-ties -- lo then belongs to another hi.
__device__ __forceinline__ _Float16 mf_opaque(_Float16 h)
{
    uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, h);
    asm volatile("" : "+v"(bits));
    return __builtin_bit_cast(_Float16, (uint16_t)bits);
}

__device__ __forceinline__ void mf_split(double v, _Float16& hi, _Float16& lo)
{
    hi = mf_opaque((_Float16)v);
    lo = (_Float16)(v - (double)hi);
}

__device__ __forceinline__ void mf_split(float v, _Float16& hi, _Float16& lo)
{
    hi = mf_opaque((_Float16)v);
    lo = (_Float16)(v - (float)hi);          // exact in f32
}

// Per-model constants of the MFMA tier.  A model (or threshold) outside the tier's domain gets s = 1, d = 0: every
// evaluation "survives" and is decided by the scalar tiers.  A slot past the end of the list gets d = 2^15, s = 0: never survives.
__device__ __forceinline__ void mf_model_setup(const double* F, bool live, float thr2, double Q1, double Q2, double P1, double P2,
                                               MfModel& o)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) { o.ch[k] = 0; o.cl[k] = 0; o.c8[k] = 0; o.s0[k] = 0; o.s1[k] = 0; }
#ifdef VO_SCORE_MF_CHECK
    o.dbg_scale = 0.; o.dbg_unit = 0.;
#endif
    if (!live) { o.c8[0] = (_Float16)32768.f; return; }       // d = 2^15, s = 0: r = -2^30
    double fm = 0., nanprobe = 0.;
#pragma unroll
    for (int k = 0; k < 9; ++k) { fm = fmax(fm, fabs(F[k])); nanprobe += F[k] * 0.; }
    bool ok = fm > 0x1p-300 && fm < 0x1p300 && thr2 >= 0x1p-60f && thr2 <= 0x1p40f && nanprobe == 0.;
    int e = 0;
    (void)frexp(ok ? fm : 1., &e);
    const double sigma = ldexp(1., -e);
    double Fs[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Fs[k] = ok ? F[k] * sigma : 0.;
    const double up = 1. + 0x1p-40;
    const double r1 = Q1 / P1 * up, r2 = Q2 / P2 * up;                 // in [0.5, 1]: bounds of the scaled coordinates
    const double pk[8] = {P2 * P1, P2 * P1, P2, P2 * P1, P2 * P1, P2, P1, P1};
    const double bk[8] = {r2 * r1, r2 * r1, r2, r2 * r1, r2 * r1, r2, r1, r1};
    double ck[8], cm = fabs(Fs[8]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { ck[k] = Fs[k] * pk[k]; cm = fmax(cm, fabs(ck[k])); }
    int ec = 0;
    (void)frexp(ok ? cm : 1., &ec);                                   // cm < 2^ec
    const double thi = (double)thr2 * (1. + 0x1p-16);
    const double P11 = P1 * P1;
    double G[5] = {(Fs[0] * Fs[0] + Fs[3] * Fs[3]) * P11, 2. * (Fs[0] * Fs[1] + Fs[3] * Fs[4]) * P11,
                   (Fs[1] * Fs[1] + Fs[4] * Fs[4]) * P11, 2. * (Fs[0] * Fs[2] + Fs[3] * Fs[5]) * P1,
                   2. * (Fs[1] * Fs[2] + Fs[4] * Fs[5]) * P1};
    const double bs[5] = {r1 * r1, r1 * r1, r1 * r1, r1, r1};
    double g1 = (Fs[2] * Fs[2] + Fs[5] * Fs[5]) * thi, T = g1, gm = 0.;
#pragma unroll
    for (int k = 0; k < 5; ++k) { G[k] *= thi; T += fabs(G[k]) * bs[k]; }
    T *= up;
    // The margin 2 eta |d| of (|d| - eta)^2 is traded for a scaling of the threshold side (one FMA less per
    // evaluation): 2 eta |d| <= dl d^2 + eta^2 / dl, so (|d| - eta)^2 >= (1 - dl) d^2 - eta^2 (1/dl - 1), and
    //   d~^2 > (thi den2~ + beta + eta^2 (1/dl - 1)) / (1 - dl)   =>   sure outlier.
    // With dl ~ eta / (thr sqrt(den2)) the band of undecided residuals is as wide as the exact margin's; den2 varies
    // over the correspondences, so dl is set from a typical value (a quarter of the bound T~), clamped to [2^-6, 1/2].
    double S0 = fabs(Fs[8]);
#pragma unroll
    for (int k = 0; k < 8; ++k) S0 += fabs(ck[k]) * bk[k];
    const double eta0 = (double)kMfEtaUnits * 0x1p-24 * S0 * up;       // eta at kappa = 1, without its absolute floor
    double dl = T > 0. ? eta0 / sqrt(0.25 * T) : 0.5;
    dl = fmin(0.5, fmax(0x1p-6, dl));
    const double fs = (1. / (1. - dl)) * up, qd = (1. / dl - 1.) * up;
#pragma unroll
    for (int k = 0; k < 5; ++k) { G[k] *= fs; gm = fmax(gm, fabs(G[k])); }
    gm = fmax(gm, ((g1 + 0x1p-17 * T) + 4. * eta0 * eta0 * qd) * fs);  // (4 eta0^2 covers eta^2 with its floor, see below)
    int eg = 0;
    (void)frexp(gm > 0. ? gm : 1., &eg);                              // gm < 2^eg
    int ek = 14 - ec;                                                 // kappa = 2^ek: max |c~| in [2^13, 2^14)
    if (eg + 2 * ek > 14) ek -= (eg + 2 * ek - 14 + 1) / 2;           // keep the den2 entries below 2^14 as well
    if (ec + ek < 3 || !(gm < 0x1p600)) ok = false;                   // ... unless that costs the d coefficients their precision
    if (!(ldexp(0x1p-20, -ek) <= eta0)) ok = false;                   // (never with |c~| >= 2^3: eta0 kappa >= 2^-18)
    if (!ok) { o.s1[7] = (_Float16)1.f; return; }
    const double kap = ldexp(1., ek), kap2 = ldexp(1., 2 * ek);
    double S = 0.;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double c = ck[k] * kap;
        _Float16 hi, lo;
        mf_split(c, hi, lo);
        o.ch[k] = hi; o.cl[k] = lo;
        S += fabs(c) * bk[k];
    }
    const double c8 = Fs[8] * kap;
    S += fabs(c8);
    {
        _Float16 hi, lo;
        mf_split(c8, hi, lo);
        o.c8[0] = hi; o.c8[1] = lo;
    }
    const double eta = ((double)kMfEtaUnits * 0x1p-24 * S + 0x1p-20) * up;
#ifdef VO_SCORE_MF_CHECK
    o.dbg_scale = sigma * kap; o.dbg_unit = 0x1p-24 * S;
#endif
    _Float16 gh[5], gl[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) mf_split(G[k] * kap2, gh[k], gl[k]);
    // constant term + beta + the eta^2 term, rounded up to f16
    const double g1t = ((g1 + 0x1p-17 * T) * kap2 + 0x1p-20 + eta * eta * qd) * fs;
    _Float16 g1h = (_Float16)g1t;
    if ((double)g1h < g1t) g1h = __builtin_bit_cast(_Float16, (uint16_t)(__builtin_bit_cast(uint16_t, g1h) + 1));
    o.s0 = v8h{gh[0], gh[1], gh[2], gh[3], gh[4], gh[0], gh[1], gh[2]};
    o.s1 = v8h{gh[3], gh[4], gl[0], gl[1], gl[2], gl[3], gl[4], g1h};
}

// Split features of one correspondence into the four point-side operand planes.  (x1 y1 x2 y2) arrive scaled by exact powers of
// two to [-1, 1]; a padding point (real = false, v = 0) gets all-zero features, the constant 1 of den2 included.
__device__ __forceinline__ void mf_stage_point(float4 v, bool real, float iP1, float iP2, uint4& dh, uint4& dl, uint4& s0, uint4& s1)
{
    const float x1 = v.x * iP1, y1 = v.y * iP1, x2 = v.z * iP2, y2 = v.w * iP2;
    const float phi[8] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1};
    const float psi[3] = {x1 * x1, x1 * y1, y1 * y1};
    v8h ph, pl;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        _Float16 hi, lo;
        mf_split(phi[k], hi, lo);
        ph[k] = hi; pl[k] = lo;
    }
    _Float16 sh[3], sl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) mf_split(psi[k], sh[k], sl[k]);
    const v8h a0 = {sh[0], sh[1], sh[2], ph[6], ph[7], sl[0], sl[1], sl[2]};
    const v8h a1 = {pl[6], pl[7], sh[0], sh[1], sh[2], ph[6], ph[7], (_Float16)(real ? 1.f : 0.f)};
    dh = __builtin_bit_cast(uint4, ph);
    dl = __builtin_bit_cast(uint4, pl);
    s0 = __builtin_bit_cast(uint4, a0);
    s1 = __builtin_bit_cast(uint4, a1);
}

// Inclusive prefix sum over the 64 lanes of a wave in registers (DPP row shifts, then the row totals broadcast across
// rows): the queue positions of a group's survivors without an LDS round trip.
__device__ __forceinline__ uint32_t mf_wave_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2, 3
    return x;
}

// two VALU instructions per evaluation: sign(r) set <=> sure outlier
__device__ __forceinline__ uint32_t mf_epilogue(const v16f& d, const v16f& s, uint32_t bits)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float rr = __builtin_fmaf(-d[r], d[r], s[r]);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(rr), 31);
    }
    return bits;
}

__global__ __launch_bounds__(kThreads, VO_SCORE_MF_WAVES)
void score_fund_mfma_kernel(const ScoreArgs a, const int gy)
{
    constexpr int C = kMfChunk, kWaves = kThreads / 64;
    constexpr int kNE = VO_SCORE_NE, kBatch = 64 * kNE;
    constexpr int kListCap = 256, kQueueCap = 64 + kBatch;
    static_assert(C >= kThreads, "the models' operands are staged through the feature planes");
    __shared__ uint4 planebuf[4 * C];                       // the correspondences' operands: d hi, d lo, den2 half 0, den2 half 1 (C x 16 B each);
    uint4 (*planes)[C] = (uint4 (*)[C])planebuf;            // first the models' operands pass through it (4 x 16 B each)
    uint4* const stagebuf = planebuf;
#ifdef VO_SCORE_MF_CHECK
    __shared__ double wdbg[kWaves][64][2];
#endif
    __shared__ uint32_t wc8[kWaves][64];                    // (c~_8 hi, lo) of every model
    __shared__ float4 spf[C];                               // the chunk's correspondences, for the scalar tiers
    __shared__ double sscale[3][kWaves];
    __shared__ uint16_t wlist[kWaves][kBatch + kListCap];  // tier-1b queue: (group of the chunk << 12) | (lane << 6) | bit of its survivor string
    __shared__ uint16_t wq2[kWaves][kQueueCap];            // tier-2 queue, same entries
    __shared__ int wcnt[kWaves][64];
    __shared__ uint32_t wslot[kWaves][64];
    __shared__ __attribute__((aligned(16))) float wflt[kWaves][64][12];

    const int b = blockIdx.z;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const int first = a.list && a.nstart ? a.nstart[b] : 0;
    const int nslots = a.list ? a.nlist[b] - first : a.H;
    const int nchunks = (n + C - 1) / C;
    if ((int)blockIdx.y >= nchunks || (int)(blockIdx.x * kThreads) >= nslots) return;      // block-uniform
    const float4* const pts = (const float4*)a.pts + (size_t)b * a.N;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, kh = lane >> 5, col = lane & 31;

    // this thread's model: the loads go out first, the bounds pass below covers their latency
    const int t = blockIdx.x * kThreads + threadIdx.x;
    int slot = t;
    bool live = t < nslots;
    if (a.list) slot = live ? a.list[(size_t)b * a.H + first + t] : 0;
    const size_t gslot = (size_t)b * a.H + slot;
    if (!a.list) live = live && (!a.valid || a.valid[gslot]);
    double M[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = live ? a.models[gslot * 9 + k] : 0.;
    // the first chunk's correspondences, two per thread, on their way while the models are set up
    constexpr int kPer = (C + kThreads - 1) / kThreads;     // correspondences of a chunk per thread
    const auto fetch = [&](int c, float4 (&v)[kPer]) {
        const int p0 = c * C, np = min(C, n - p0);
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = threadIdx.x + u * kThreads;
            v[u] = i < np ? pts[p0 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 nxt[kPer];
    fetch(blockIdx.y, nxt);

    // problem-wide bounds on max(|x|,|y|,1) of either image (the models' constants are set up once for all chunks);
    // f32 maxima of f32 values are exact, four loads in flight per thread
    if (a.bounds) {                                          // block-uniform: found once per problem by vo_launch_fund_bounds
        const float4 pb = *(const float4*)(a.bounds + 4 * (size_t)b);
        if (lane == 0) { sscale[0][w] = (double)pb.x; sscale[1][w] = (double)pb.y; sscale[2][w] = (double)pb.z; }
    } else {
        float q1 = 1.f, q2 = 1.f;
        bool tame = true;
        for (int i = threadIdx.x; i < n; i += 4 * kThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = i + u * kThreads < n ? pts[i + u * kThreads] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q1 = fmaxf(q1, fmaxf(fabsf(v[u].x), fabsf(v[u].y)));      // a NaN is dropped here and caught by `tame`
                q2 = fmaxf(q2, fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
                tame = tame && fabsf(v[u].x) <= 0x1p20f && fabsf(v[u].y) <= 0x1p20f && fabsf(v[u].z) <= 0x1p20f && fabsf(v[u].w) <= 0x1p20f;
            }
        }
        for (int o = 32; o; o >>= 1) {
            q1 = fmaxf(q1, __shfl_xor(q1, o));
            q2 = fmaxf(q2, __shfl_xor(q2, o));
        }
        tame = __all(tame);
        if (lane == 0) { sscale[0][w] = (double)q1; sscale[1][w] = (double)q2; sscale[2][w] = tame ? 1. : 0.; }
    }
    __syncthreads();
    double p1 = sscale[0][0], p2 = sscale[1][0], tame_all = sscale[2][0];
#pragma unroll
    for (int q = 1; q < kWaves; ++q) {
        p1 = fmax(p1, sscale[0][q]);
        p2 = fmax(p2, sscale[1][q]);
        tame_all = fmin(tame_all, sscale[2][q]);
    }
    const bool fast = tame_all != 0.;                       // workgroup-uniform
    int e1 = 0, e2 = 0;
    (void)frexp(p1, &e1);
    (void)frexp(p2, &e2);
    const double P1 = ldexp(1., e1), P2 = ldexp(1., e2);   // powers of two >= the bounds
    const float iP1 = (float)ldexp(1., -e1), iP2 = (float)ldexp(1., -e2);

    // per-model constants: the scalar tiers' f32 filter (LDS, read by whichever lane works a queued pair) and the MFMA
    // tier's model-side operands (staged through the feature planes, which nothing has written yet)
    float tlo, thi;                                         // functions of the threshold alone
    {
        FundFilter32 f32;
        f32 = fund_filter32_setup(M, a.thr2, p1, p2);
        float4* o = (float4*)wflt[w][lane];
        o[0] = make_float4(f32.F[0], f32.F[1], f32.F[2], f32.F[3]);
        o[1] = make_float4(f32.F[4], f32.F[5], f32.F[6], f32.F[7]);
        o[2] = make_float4(f32.F[8], f32.eta2, f32.c, 0.f);
        wcnt[w][lane] = 0;
        wslot[w][lane] = (uint32_t)gslot;
        if (fast) {
            MfModel mm;
            mf_model_setup(M, live, a.thr2, p1, p2, P1, P2, mm);
            uint4* st = &stagebuf[(size_t)threadIdx.x * 4];
            st[0] = __builtin_bit_cast(uint4, mm.ch);
            st[1] = __builtin_bit_cast(uint4, mm.cl);
            st[2] = __builtin_bit_cast(uint4, mm.s0);
            st[3] = __builtin_bit_cast(uint4, mm.s1);
            wc8[w][lane] = __builtin_bit_cast(uint4, mm.c8).x;
#ifdef VO_SCORE_MF_CHECK
            wdbg[w][lane][0] = mm.dbg_scale; wdbg[w][lane][1] = mm.dbg_unit;
#endif
        }
        tlo = f32.tlo; thi = f32.thi;
    }
    __syncthreads();
    // (a wave past the end of the model list stays: it helps staging the chunks and keeps the barriers; its models are
    // "never survive" columns)
    // (names from before the tile was transposed: Bd0 / Bd1 / Bs hold the MODELS' vectors and are the MFMA's first, row-side
    // operand; ad / as below are the correspondences' and its second, column-side one)
    v8h Bd0[2], Bd1[2], Bs[2];
    if (fast) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int m = 64 * w + 32 * cb + col;
            const uint4* st = &stagebuf[(size_t)m * 4];
            Bd0[cb] = __builtin_bit_cast(v8h, st[0]);
            // lanes 0-31: lo halves against phi hi; 32-63: c~_8 against (1, 1, 0...)
            Bd1[cb] = __builtin_bit_cast(v8h, kh ? make_uint4(wc8[w][32 * cb + col], 0u, 0u, 0u) : st[1]);
            Bs[cb] = __builtin_bit_cast(v8h, st[2 + kh]);
        }
    }

    const auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // the scalar tiers of score_kernel<kFund>, with the model of a queued pair looked up by its index in the wave
    const auto load_filter = [&](int m) {
        FundFilter32 g;
        const float4* fm = (const float4*)wflt[w][m];
        const float4 A = fm[0], B = fm[1], Cc = fm[2];
        g.F[0] = A.x; g.F[1] = A.y; g.F[2] = A.z; g.F[3] = A.w; g.F[4] = B.x; g.F[5] = B.y; g.F[6] = B.z; g.F[7] = B.w;
        g.F[8] = Cc.x; g.eta2 = Cc.y; g.c = Cc.z;
        g.thi = thi; g.tlo = tlo;
        return g;
    };
    uint32_t q2 = 0;                                         // wave-uniform: entries waiting in the tier-2 queue
    const auto flush2 = [&](uint32_t take, uint32_t base) {
        if ((uint32_t)lane < take) {
            const uint32_t ent = wq2[w][base + lane];
            const int m = ent >> 9, i = ent & 511;
            const float4 v = spf[i];
            if (fund_decide64(a.models + (size_t)wslot[w][m] * 9, v.x, v.y, v.z, v.w, a.thr2, p1, p2)) atomicAdd(&wcnt[w][m], 1);
#ifdef VO_SCORE_STATS
            atomicAdd(&g_stats[2], 1ull);
#endif
        }
    };
    const auto flush1 = [&](uint32_t take, uint32_t base) {
        bool has[kNE];
        uint32_t ent[kNE];
#pragma unroll
        for (int j = 0; j < kNE; ++j) {
            has[j] = (uint32_t)(lane + 64 * j) < take;
            const uint32_t raw = has[j] ? wlist[w][base + lane + 64 * j] : 0u;
            // (group, lane, bit) -> (model of the wave << 9) | correspondence of the chunk
            const int bit = raw & 63, sl = (raw >> 6) & 63, nn = 31 - (bit & 31);
            const int m = 32 * (nn >> 4) + 8 * ((nn & 15) >> 2) + 4 * (sl >> 5) + (nn & 3);
            const int pt = 64 * (int)(raw >> 12) + 32 * (1 - (bit >> 5)) + (sl & 31);
            ent[j] = (uint32_t)((m << 9) | pt);
        }
        int verdict[kNE];
#pragma unroll
        for (int j = 0; j < kNE; ++j) {
            const FundFilter32 g = load_filter(ent[j] >> 9);
            const float4 v = spf[ent[j] & 511];
            verdict[j] = fund_filter32(g, v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (int j = 0; j < kNE; ++j) {
            const bool undecided = has[j] && verdict[j] == 0;
            const unsigned long long um = __ballot(undecided);               // tier-2 queue positions from the ballot: no LDS counter
            if (has[j] && verdict[j] > 0) atomicAdd(&wcnt[w][ent[j] >> 9], 1);
            if (undecided) wq2[w][q2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u))] = (uint16_t)ent[j];
            q2 += (uint32_t)__popcll(um);
        }
#ifdef VO_SCORE_STATS
        if (lane == 0) atomicAdd(&g_stats[1], 1ull);
#endif
    };
    const auto drain2 = [&](uint32_t keep) {
        if (q2 > keep) {
            wave_sync();
            do {
                const uint32_t take = min(q2, 64u);
                q2 -= take;
                flush2(take, q2);
            } while (q2 > keep);
            wave_sync();
        }
    };
    // survivor bit k of a (lane, column block) string over one group of 64 correspondences: evaluation n = 31 - k in
    // issue order = row block n >> 4, accumulator register n & 15
    const auto model_at = [&](int n) { return 32 * (n >> 4) + 8 * ((n & 15) >> 2) + 4 * kh + (n & 3); };

#ifdef VO_SCORE_MF_PROF
    long long prof[6] = {0, 0, 0, 0, 0, 0};
    const long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int c = blockIdx.y; c < nchunks; c += gy) {
        const int p0 = c * C, np = min(C, n - p0);
        MF_T(ts0);
        __syncthreads();                                     // the previous chunk (or the model staging) has been read
        {
            const int npad = (np + 63) & ~63;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = threadIdx.x + u * kThreads;
                if (i < npad) {
                    spf[i] = nxt[u];
                    if (fast) mf_stage_point(nxt[u], i < np, iP1, iP2, planes[0][i], planes[1][i], planes[2][i], planes[3][i]);
                }
            }
        }
        __syncthreads();
        if (c + gy < nchunks) fetch(c + gy, nxt);            // the next chunk's loads fly under this chunk's work
        MF_T(ts1);
        MF_ACC(0, ts0, ts1);
        if (!fast) {
            // a problem with non-finite or huge coordinates: the owner lane's compare-form filter, then tier 2 by the owner
            int cnt = 0;
            const FundFilter32 f32 = load_filter(lane);
            for (int i0 = 0; i0 < np; i0 += 32) {
                const int nb = min(32, np - i0);
                uint32_t bits = 0;
#pragma unroll 1
                for (int k = 0; k < nb; ++k) {
                    const float4 v = spf[i0 + k];
                    const int verdict = fund_filter32(f32, v.x, v.y, v.z, v.w);
                    cnt += verdict > 0;
                    bits = (bits << 1) | (uint32_t)(verdict == 0);
                }
                if (!live) bits = 0;
                while (__any(bits != 0)) {
                    if (bits) {
                        const int pos = 31 - __clz(bits);
                        bits &= ~(1u << pos);
                        const float4 v = spf[i0 + nb - 1 - pos];
                        cnt += fund_decide64(a.models + gslot * 9, v.x, v.y, v.z, v.w, a.thr2, p1, p2);
                    }
                }
            }
            if (live && cnt) atomicAdd(&wcnt[w][lane], cnt);
            continue;
        }

        const int ngroups = (np + 63) >> 6;
        uint32_t keep = kh ? 0u : ~0u, ones = kh ? 0x3c003c00u : 0u;    // lanes 32-63: (1, 1, 0...) instead of phi hi
        asm volatile("" : "+v"(keep), "+v"(ones));
        uint32_t q1 = 0;                                     // wave-uniform: entries waiting in the tier-1b queue
        const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < ngroups; ++g) {
            const int i0 = 64 * g;
            MF_T(ta0);
            v8h ad[2], as[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ad[h] = __builtin_bit_cast(v8h, planes[kh][i0 + 32 * h + col]);
                as[h] = __builtin_bit_cast(v8h, planes[2 + kh][i0 + 32 * h + col]);
            }
            uint32_t bits[2] = {0u, 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) {                   // tile q of the group: correspondence block q >> 1, model block q & 1
                const int h = q >> 1, cb = q & 1;
                // second K step of d: lanes 0-31 pair phi hi with the lo coefficients, lanes 32-63 pair (1, 1, 0...) with c~_8
                const uint4 adw = __builtin_bit_cast(uint4, ad[h]);       // (bit masks: a v_cndmask_b32 costs 12 cycles, a v_and_b32 2)
                const v8h a1 = __builtin_bit_cast(v8h, make_uint4((adw.x & keep) | ones, adw.y & keep, adw.z & keep, adw.w & keep));
                // models along the rows, correspondences along the columns: a lane's 16 accumulators are 16 MODELS for
                // one correspondence (column = lane & 31), which spreads the survivors of a good model over the lanes
                v16f accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bd0[cb], ad[h], zero16, 0, 0, 0);
                accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bd1[cb], a1, accd, 0, 0, 0);
                const v16f accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bs[cb], as[h], zero16, 0, 0, 0);
#ifdef VO_SCORE_MF_CHECK
                // measurement build: the MFMA's d~ against the f64 value of kappa x2^T F x1, in budget units (eta = 48 of them)
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * cb + 8 * (r >> 2) + 4 * kh + (r & 3), pt = i0 + 32 * h + col;
                    const double unit = wdbg[w][m][1];
                    if (pt < np && unit > 0.) {
                        const double* F = a.models + (size_t)wslot[w][m] * 9;
                        const float4 v = spf[pt];
                        const double x1 = v.x, y1 = v.y, x2 = v.z, y2 = v.w;
                        const double aa = __builtin_fma(F[0], x1, __builtin_fma(F[1], y1, F[2])), bb = __builtin_fma(F[3], x1, __builtin_fma(F[4], y1, F[5])),
                                     cc = __builtin_fma(F[6], x1, __builtin_fma(F[7], y1, F[8]));
                        const double dd = __builtin_fma(x2, aa, __builtin_fma(y2, bb, cc)) * wdbg[w][m][0];
                        const double units = fabs((double)accd[r] - dd) / unit;
                        atomicMax(&g_stats[0], (unsigned long long)(units * 1000.));
                        atomicAdd(&g_stats[1], 1ull);
                        if (units > 4.) atomicAdd(&g_stats[2], 1ull);
                    }
                }
#endif
                bits[h] = mf_epilogue(accd, accs, bits[h]);
            }
            // bit 32 (1 - h) + 31 - n: correspondence i0 + 32 h + column, model 32 cb + row of accumulator r, n = 16 cb + r
            unsigned long long cand = ~(((unsigned long long)bits[0] << 32) | bits[1]);
            if (i0 + 64 > np)                                // ragged end of the chunk: padding points decide nothing
                cand &= (i0 + col < np ? 0xffffffff00000000ull : 0ull) | (i0 + 32 + col < np ? 0xffffffffull : 0ull);
            MF_T(ta1);
            MF_ACC(1, ta0, ta1);
            const uint32_t mine = (uint32_t)__popcll(cand), incl = mf_wave_scan(mine);
            uint32_t at = q1 + incl - mine;                  // this lane's place in the queue
            const uint32_t q1new = q1 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#ifdef VO_SCORE_STATS
            atomicAdd(&g_stats[0], (unsigned long long)__popcll(cand));
            if (lane == 0) atomicAdd(&g_stats[3], 1ull);
#endif
            if (q1new - q1 <= (uint32_t)kListCap) {
                // raw entries (group << 12) | (lane << 6) | bit: this loop runs as long as the wave's busiest lane has
                // survivors, so it only finds and stores; flush1 decodes (there every lane has work)
                uint32_t c0 = (uint32_t)(cand >> 32), c1 = (uint32_t)cand;
                const uint32_t tag = (uint32_t)(g << 12) | (uint32_t)(lane << 6);
                while (c0) {
                    wlist[w][at++] = (uint16_t)(tag | 32u | (uint32_t)__builtin_ctz(c0));
                    c0 &= c0 - 1;
                }
                while (c1) {
                    wlist[w][at++] = (uint16_t)(tag | (uint32_t)__builtin_ctz(c1));
                    c1 &= c1 - 1;
                }
                q1 = q1new;
                wave_sync();
                MF_T(tb1);
                MF_ACC(2, ta1, tb1);
                if (q1 >= (uint32_t)kBatch) {
                    do {
                        q1 -= kBatch;
                        MF_T(tc0);
                        flush1(kBatch, q1);
                        wave_sync();
                        MF_T(tc1);
                        MF_ACC(3, tc0, tc1);
                        drain2(63);
                        MF_T(tc2);
                        MF_ACC(4, tc1, tc2);
                    } while (q1 >= (uint32_t)kBatch);
                }
            } else {
                // more survivors in one group than the queue has room for: every lane works its own strings off
                while (__any(cand != 0)) {
                    if (cand) {
                        const int bit = __ffsll((long long)cand) - 1;
                        cand &= cand - 1;
                        const int m = model_at(31 - (bit & 31));
                        const float4 v = spf[i0 + 32 * (1 - (bit >> 5)) + col];
                        const FundFilter32 gf = load_filter(m);
                        int verdict = fund_filter32(gf, v.x, v.y, v.z, v.w);
                        if (verdict == 0)
                            verdict = fund_decide64(a.models + (size_t)wslot[w][m] * 9, v.x, v.y, v.z, v.w, a.thr2, p1, p2) ? 1 : -1;
                        if (verdict > 0) atomicAdd(&wcnt[w][m], 1);
                    }
                }
            }
        }
        // the queues index this chunk's correspondences: empty them before the next chunk is staged
        MF_T(te0);
        wave_sync();
        if (q1) flush1(q1, 0);
        wave_sync();
        drain2(0);
        wave_sync();
        MF_T(te1);
        MF_ACC(5, te0, te1);
    }
#ifdef VO_SCORE_MF_PROF
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&g_stats[k], (unsigned long long)prof[k]);
        atomicAdd(&g_stats[6], (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
        atomicAdd(&g_stats[7], 1ull);
    }
#endif
    wave_sync();
    const int cnt = lds_load(&wcnt[w][lane]);
    if (live && cnt) atomicAdd(a.counts + gslot, cnt);
}

// ---- upper bounds instead of counts: the first tier alone ------------------------------------------------------------
// The evaluations the matrix-core tier cannot prove to be outliers are a superset of the inliers, so
//   ub(model) = correspondences - proven outliers  >=  the model's inlier count.
// Inside a RANSAC run that is all most hypotheses need: a model whose upper bound does not exceed the best count of the
// iterations already replayed can never be a strict improvement (OpenCV's loop only acts on good > maxGoodCount), so its
// survivors need no second tier.  screen_fund_mfma_kernel is score_fund_mfma_kernel without queues and scalar tiers; the
// tile is transposed back (correspondences along the rows, models along the columns: a lane's 16 accumulators are 16
// correspondences of ONE model, so its proven outliers are a popcount of its own bit string).  It ADDS the proven
// outliers of its chunks to counts[slot] (zero before); screen_finalize_kernel turns them into upper bounds and lists the
// contenders, which score_fund_mfma_kernel then counts exactly (vo_launch_score_fund_pruned in vo_score.hip).
// A problem with non-finite or huge coordinates proves nothing (every model a contender); so does a model outside the
// tier's domain (mf_model_setup: s = 1, d = 0).
#ifndef VO_SCREEN_CHUNK
#define VO_SCREEN_CHUNK 256
#endif
#ifndef VO_SCREEN_WAVES
#define VO_SCREEN_WAVES 4
#endif
#ifndef VO_SCREEN_MB
#define VO_SCREEN_MB 2               // blocks of 256 models per workgroup: the staged correspondences serve both
#endif
__global__ __launch_bounds__(kThreads, VO_SCREEN_WAVES)
void screen_fund_mfma_kernel(const ScoreArgs a, const int gy)
{
    constexpr int C = VO_SCREEN_CHUNK, kWaves = kThreads / 64, kMB = VO_SCREEN_MB;       // kMB blocks of 256 models per workgroup
    static_assert(C >= kThreads, "the models' operands are staged through the feature planes");
    __shared__ uint4 planebuf[4 * C];
    uint4 (*planes)[C] = (uint4 (*)[C])planebuf;
    uint4* const stagebuf = planebuf;
    __shared__ uint32_t wc8[kWaves][64];
    __shared__ uint32_t wslot[kMB][kWaves][64];             // global slot of every model, ~0u past the end of the list
    __shared__ double sscale[3][kWaves];

    const int b = blockIdx.z;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const int first = a.list && a.nstart ? a.nstart[b] : 0;
    const int nslots = a.list ? a.nlist[b] - first : a.H;
    const int nchunks = (n + C - 1) / C;
    if ((int)blockIdx.y >= nchunks || (int)(blockIdx.x * kThreads * kMB) >= nslots) return;      // block-uniform
    const float4* const pts = (const float4*)a.pts + (size_t)b * a.N;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, kh = lane >> 5, col = lane & 31;

    // this thread's kMB models (one per block of 256): the loads go out first
    size_t gslot[kMB];
    bool live[kMB];
    double M[kMB][9];
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb) {
        const int t = (blockIdx.x * kMB + mb) * kThreads + threadIdx.x;
        int slot = t;
        live[mb] = t < nslots;
        if (a.list) slot = live[mb] ? a.list[(size_t)b * a.H + first + t] : 0;
        gslot[mb] = (size_t)b * a.H + slot;
        if (!a.list) live[mb] = live[mb] && (!a.valid || a.valid[gslot[mb]]);
#pragma unroll
        for (int k = 0; k < 9; ++k) M[mb][k] = live[mb] ? a.models[gslot[mb] * 9 + k] : 0.;
    }
    constexpr int kPer = (C + kThreads - 1) / kThreads;
    const auto fetch = [&](int c, float4 (&v)[kPer]) {
        const int p0 = c * C, np = min(C, n - p0);
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = threadIdx.x + u * kThreads;
            v[u] = i < np ? pts[p0 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 nxt[kPer];
    fetch(blockIdx.y, nxt);

    if (a.bounds) {
        const float4 pb = *(const float4*)(a.bounds + 4 * (size_t)b);
        if (lane == 0) { sscale[0][w] = (double)pb.x; sscale[1][w] = (double)pb.y; sscale[2][w] = (double)pb.z; }
    } else {
        float q1 = 1.f, q2 = 1.f;
        bool tame = true;
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const float4 v = pts[i];
            q1 = fmaxf(q1, fmaxf(fabsf(v.x), fabsf(v.y)));
            q2 = fmaxf(q2, fmaxf(fabsf(v.z), fabsf(v.w)));
            tame = tame && fabsf(v.x) <= 0x1p20f && fabsf(v.y) <= 0x1p20f && fabsf(v.z) <= 0x1p20f && fabsf(v.w) <= 0x1p20f;
        }
        for (int o = 32; o; o >>= 1) {
            q1 = fmaxf(q1, __shfl_xor(q1, o));
            q2 = fmaxf(q2, __shfl_xor(q2, o));
        }
        tame = __all(tame);
        if (lane == 0) { sscale[0][w] = (double)q1; sscale[1][w] = (double)q2; sscale[2][w] = tame ? 1. : 0.; }
    }
    __syncthreads();
    double p1 = sscale[0][0], p2 = sscale[1][0], tame_all = sscale[2][0];
#pragma unroll
    for (int q = 1; q < kWaves; ++q) {
        p1 = fmax(p1, sscale[0][q]);
        p2 = fmax(p2, sscale[1][q]);
        tame_all = fmin(tame_all, sscale[2][q]);
    }
    if (tame_all == 0.) return;                             // workgroup-uniform: nothing proven, every model stays a contender
    int e1 = 0, e2 = 0;
    (void)frexp(p1, &e1);
    (void)frexp(p2, &e2);
    const double P1 = ldexp(1., e1), P2 = ldexp(1., e2);
    const float iP1 = (float)ldexp(1., -e1), iP2 = (float)ldexp(1., -e2);
    v8h Md0[kMB][2], Md1[kMB][2], Ms[kMB][2];                // the models' vectors: the MFMA's column-side operand
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb) {
        if (mb) __syncthreads();                             // the staging buffer has been read
        {
            MfModel mm;
            mf_model_setup(M[mb], live[mb], a.thr2, p1, p2, P1, P2, mm);
            uint4* st = &stagebuf[(size_t)threadIdx.x * 4];
            st[0] = __builtin_bit_cast(uint4, mm.ch);
            st[1] = __builtin_bit_cast(uint4, mm.cl);
            st[2] = __builtin_bit_cast(uint4, mm.s0);
            st[3] = __builtin_bit_cast(uint4, mm.s1);
            wc8[w][lane] = __builtin_bit_cast(uint4, mm.c8).x;
            wslot[mb][w][lane] = live[mb] ? (uint32_t)gslot[mb] : ~0u;
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int m = 64 * w + 32 * cb + col;
            const uint4* st = &stagebuf[(size_t)m * 4];
            Md0[mb][cb] = __builtin_bit_cast(v8h, st[0]);
            Md1[mb][cb] = __builtin_bit_cast(v8h, kh ? make_uint4(wc8[w][32 * cb + col], 0u, 0u, 0u) : st[1]);
            Ms[mb][cb] = __builtin_bit_cast(v8h, st[2 + kh]);
        }
    }
    uint32_t keep = kh ? 0u : ~0u, ones = kh ? 0x3c003c00u : 0u;
    asm volatile("" : "+v"(keep), "+v"(ones));
    const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t outl[kMB][2];                                  // proven outliers of models 32 cb + col (block mb) among this lane half's rows
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb) outl[mb][0] = outl[mb][1] = 0u;
    for (int c = blockIdx.y; c < nchunks; c += gy) {
        const int p0 = c * C, np = min(C, n - p0);
        __syncthreads();                                     // the previous chunk (or the model staging) has been read
        {
            const int npad = (np + 63) & ~63;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = threadIdx.x + u * kThreads;
                if (i < npad) mf_stage_point(nxt[u], i < np, iP1, iP2, planes[0][i], planes[1][i], planes[2][i], planes[3][i]);
            }
        }
        __syncthreads();
        if (c + gy < nchunks) fetch(c + gy, nxt);
        const int ngroups = (np + 63) >> 6;
        for (int g = 0; g < ngroups; ++g) {
            const int i0 = 64 * g;
            v8h pd[2], ps[2];                                // the correspondences' vectors: the row-side operand
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pd[h] = __builtin_bit_cast(v8h, planes[kh][i0 + 32 * h + col]);
                ps[h] = __builtin_bit_cast(v8h, planes[2 + kh][i0 + 32 * h + col]);
            }
            // bit 31 - 16 h - r of a string: correspondence i0 + 32 h + 8 (r >> 2) + 4 kh + (r & 3).  A padding row has zero
            // features but still meets the constant term c~_8: it is masked out of the ragged last group.
            uint32_t vmask = ~0u;
            if (i0 + 64 > np) {
                vmask = 0u;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int h = k >> 4, r = k & 15;
                    if (i0 + 32 * h + 8 * (r >> 2) + 4 * kh + (r & 3) < np) vmask |= 1u << (31 - k);
                }
            }
            v8h p1v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 pdw = __builtin_bit_cast(uint4, pd[h]);
                p1v[h] = __builtin_bit_cast(v8h, make_uint4((pdw.x & keep) | ones, pdw.y & keep, pdw.z & keep, pdw.w & keep));
            }
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) {
                uint32_t bits[2] = {0u, 0u};
#pragma unroll
                for (int q = 0; q < 4; ++q) {               // tile q: correspondence block q >> 1, model block q & 1
                    const int h = q >> 1, cb = q & 1;
                    v16f accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(pd[h], Md0[mb][cb], zero16, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1v[h], Md1[mb][cb], accd, 0, 0, 0);
                    const v16f accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ps[h], Ms[mb][cb], zero16, 0, 0, 0);
                    bits[cb] = mf_epilogue(accd, accs, bits[cb]);
                }
                outl[mb][0] += (uint32_t)__popc(bits[0] & vmask);
                outl[mb][1] += (uint32_t)__popc(bits[1] & vmask);
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const uint32_t tot = outl[mb][cb] + (uint32_t)__shfl_xor((int)outl[mb][cb], 32);
            const uint32_t gs = wslot[mb][w][32 * cb + col];
            if (kh == 0 && gs != ~0u && tot) atomicAdd(a.counts + gs, (int)tot);
        }
}

// counts[slot] holds the proven outliers of list entries [nstart, nlist) of every problem: turn them into upper bounds;
// a model whose bound exceeds the best count so far (result[4 b + 2]) goes to the contender list (length in ncl[b * kNclStride]: one counter per 64 bytes -- adjacent counters would share
// cache lines and their atomics an L2 channel, 27 vs 76 us for this kernel); counts[slot] keeps the bound until count_contenders_kernel replaces it
constexpr int kNclStride = 16;
__global__ __launch_bounds__(256)
void screen_finalize_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ nstart, const int32_t* __restrict__ nlist,
                            int Hs, const int32_t* __restrict__ n_dev, int N, const int32_t* __restrict__ result,
                            int32_t* __restrict__ counts, int32_t* __restrict__ clist, int32_t* __restrict__ ncl)
{
    const int b = blockIdx.y;
    const int first = nstart ? nstart[b] : 0, cnt = nlist[b] - first;
    const int n = n_dev ? min(n_dev[b], N) : N;
    const int best = result[4 * b + 2];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < cnt; t += gridDim.x * blockDim.x) {
        const int slot = list[(size_t)b * Hs + first + t];
        const size_t gs = (size_t)b * Hs + slot;
        const int ub = n - counts[gs];
        counts[gs] = ub;
        if (ub > best) clist[(size_t)b * Hs + atomicAdd(&ncl[(size_t)b * kNclStride], 1)] = slot;
    }
}
