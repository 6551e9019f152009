// vo_knn.hip -- brute-force 2-nearest-neighbour search over 128-D SIFT descriptors on
// gfx950 int8 MFMA.  Stands in for cv.BFMatcher().knnMatch(des_i, des_j, k=2) at reference
// src/features.py:100-101 (semantics: SURVEY.md Appendix A.1).
//
// Arithmetic: descriptors are integers 0..255 stored as int8 (value - 128).  For one query
// the order of squared distances D = |q|^2 + |t|^2 - 2 q.t over train rows is the reverse
// order of key = 2 q.t - |t|^2, so each lane maximises key; q.t comes out of
// v_mfma_i32_32x32x32_i8 exactly (int32 accumulate), and D is rebuilt at the very end.
//
// Data movement: a workgroup (8 waves) owns 512 query rows (2 sets of 32 per wave) and one split of the train
// rows.  Every wave keeps its query rows as MFMA B fragments in registers for the whole sweep (queries sit on
// the MFMA column = lane axis, so a lane's running top-2 never leaves the lane).  Train rows stream through LDS
// in 128-row chunks (LDS-DMA, double buffered, 16-byte slots XOR-swizzled on the source side so that the
// ds_read_b128 of an A fragment is bank-conflict free) and are shared by the eight waves.  Per 32x32 output tile
// the VALU does 2.5 ops per element: the packed key
//     p  = (dot << 9) + c[row]        c[row] = (-|t|^2 << 8) | (15-tile)<<4 | (15-slot)
// and, two keys at a time, m1 = max3(m1, p0, p1), m2 = max(m2, med3(m1, p0, p1)).
// The low 8 bits make p unique inside a 16-tile segment and order equal keys by ascending train index, which
// is OpenCV's tie rule (strict '<' insertion while scanning upwards).  Workgroups of one problem are pinned to
// one XCD (VO_KNN_XCD) so that its train rows are fetched into one L2 only.
#include "vo_common.h"
#include <limits.h>
#include <stdlib.h>

// shape of the sweep kernel (defaults chosen by measurement, scripts/knn_variants.sh)
#ifndef VO_KNN_SETS
#define VO_KNN_SETS 2        /* 32-query sets per wave */
#endif
#ifndef VO_KNN_WAVES
#define VO_KNN_WAVES 8       /* waves per workgroup */
#endif
#ifndef VO_KNN_TILES
#define VO_KNN_TILES 4       /* 32-row train tiles per LDS stage */
#endif
#ifndef VO_KNN_PIPE
#define VO_KNN_PIPE 0        /* 1: MFMAs of the next tile issued ahead of the current tile's epilogue (measured: no gain) */
#endif
#ifndef VO_KNN_CHAINS
#define VO_KNN_CHAINS 2      /* independent top-2 lists per query set in the epilogue (1, 2 or 4) */
#endif
#ifndef VO_KNN_MINW
#define VO_KNN_MINW 2        /* waves per SIMD the register allocation must allow */
#endif
#ifndef VO_SCHED
#define VO_SCHED 0
#endif
#ifndef VO_KNN_XCD
#define VO_KNN_XCD 1         /* problems of a batch are pinned to XCDs (shared L2 for the train rows of a problem) */
#endif

namespace {

constexpr int kDim = VO_DESC_DIM;
constexpr int kTileRows = 32;
constexpr int kChunkTiles = VO_KNN_TILES;
constexpr int kChunkRows = kTileRows * kChunkTiles;          // 128 train rows per LDS stage
constexpr int kSegTiles = 16;                                 // tile index bits in the key
constexpr int kSets = VO_KNN_SETS;                                      // 32-query sets per wave
constexpr int kChains = VO_KNN_CHAINS;
constexpr int kWaveQ = 32 * kSets;                            // query rows per wave
constexpr int kWgWaves = VO_KNN_WAVES;
constexpr int kWgQ = kWaveQ * kWgWaves;                       // 256 query rows per workgroup
constexpr int kWgThreads = 64 * kWgWaves;
constexpr int kBufBytes = kChunkRows * kDim + kChunkRows * 4; // rows + their c[] words
constexpr int kLdsBytes = 2 * kBufBytes;

struct Top2 {
    int k1, j1, k2, j2;   // keys descending; j = -1 marks an empty slot (key INT_MIN)
};

__device__ __forceinline__ bool better(int ka, int ja, int kb, int jb)
{
    return ka > kb || (ka == kb && (unsigned)ja < (unsigned)jb);   // j=-1 -> 0xffffffff loses
}

// candidate arrives in ascending-j order relative to what is already held
__device__ __forceinline__ void insert_later(Top2& r, int k, int j)
{
    if (j < 0) return;
    if (k > r.k1) { r.k2 = r.k1; r.j2 = r.j1; r.k1 = k; r.j1 = j; }
    else if (k > r.k2) { r.k2 = k; r.j2 = j; }
}

__device__ __forceinline__ Top2 merge2(const Top2& a, const Top2& b)
{
    Top2 o;
    if (better(a.k1, a.j1, b.k1, b.j1)) {
        o.k1 = a.k1; o.j1 = a.j1;
        if (better(a.k2, a.j2, b.k1, b.j1)) { o.k2 = a.k2; o.j2 = a.j2; }
        else { o.k2 = b.k1; o.j2 = b.j1; }
    } else {
        o.k1 = b.k1; o.j1 = b.j1;
        if (better(a.k1, a.j1, b.k2, b.j2)) { o.k2 = a.k1; o.j2 = a.j1; }
        else { o.k2 = b.k2; o.j2 = b.j2; }
    }
    return o;
}

// v_med3_i32 as inline asm: besides the missing builtin, the asm statements keep hipcc's scheduler
// from hoisting every MFMA and LDS read of the unrolled chunk to the top (which spills); the
// source order below is the intended issue order
__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;
    asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int max3i_op(int a, int b, int c)
{
    int r;
    asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef __attribute__((address_space(3))) void lds_void;

// One 1 KiB LDS-DMA piece = 8 train rows: lane L fills the 16-byte slot (row L>>3, slot L&7) of
// the piece.  The destination of an LDS-DMA is lane-linear, so the XOR swizzle of the image goes
// on the SOURCE: slot pc of row R holds the row's 16-byte column pc ^ ((R>>1)&7).
__device__ __forceinline__ void stage_piece(const int8_t* __restrict__ t, char* lds_piece, int row_in_chunk0,
                                            int grow0, int row_end, int lane)
{
    const int r = lane >> 3, pc = lane & 7;
    const int R = row_in_chunk0 + r, gr = grow0 + r;
    if (gr < row_end) {
        const int ck = pc ^ ((R >> 1) & 7);
        __builtin_amdgcn_global_load_lds((const void*)(t + (size_t)gr * kDim + ck * 16), (lds_void*)lds_piece, 16, 0, 0);
    } else {
        const v4i z = {0, 0, 0, 0};                       // rows past the split: all-zero => dot 0
        *(v4i*)(lds_piece + lane * 16) = z;
    }
}

__global__ __launch_bounds__(kWgThreads, VO_KNN_MINW)
void knn2_partial_kernel(const int8_t* __restrict__ q, const int8_t* __restrict__ t,
                         const int32_t* __restrict__ tnorm, int nq, int nt,
                         int rows_per_split, int nsplit, int4* __restrict__ partial,
                         const int32_t* __restrict__ qnorm, int32_t* __restrict__ idx,
                         int32_t* __restrict__ d2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object (2 buffers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt to the 8 XCDs round robin in launch order (x fastest).  All workgroups of one
    // problem stream the same train rows, so they should share an L2: with the batch a multiple of 8, the
    // workgroups an XCD receives are re-labelled to walk through whole problems one after the other.
    int qblock = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
#if VO_KNN_XCD
    if ((gridDim.z & 7) == 0) {
        const int per = gridDim.x * gridDim.y;
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int xcd = lin & 7, slot = lin >> 3;
        const int inner = slot % per;
        b = (slot / per) * 8 + xcd;
        qblock = inner % gridDim.x;
        split = inner / gridDim.x;
    }
#endif
    q += (size_t)b * nq * kDim;
    t += (size_t)b * nt * kDim;
    tnorm += (size_t)b * nt;

    const int row0 = split * rows_per_split;
    const int row_end = min(nt, row0 + rows_per_split);
    const int nchunks = row_end > row0 ? (row_end - row0 + kChunkRows - 1) / kChunkRows : 0;

    // ---- query rows -> MFMA B fragments (lane: column = query lane&31, k-slice by lane>>5)
    const int qrow = lane & 31, h = lane >> 5;
    const int qbase = qblock * kWgQ + wave * kWaveQ;
    v4i qb[kSets][4];
#pragma unroll
    for (int set = 0; set < kSets; ++set) {
        const int qi = qbase + set * 32 + qrow;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4i v = *(const v4i*)(q + (size_t)min(qi, nq - 1) * kDim + 32 * s + 16 * h);
            const v4i z = {0, 0, 0, 0};
            qb[set][s] = qi < nq ? v : z;
        }
    }

    // ---- staging: every wave DMAs 8 pieces (64 rows) of a chunk; thread tid forms the key
    // constant of row tid
    auto stage_rows = [&](int chunk, int buf) {
        char* base = smem + buf * kBufBytes;
        const int rbase = row0 + chunk * kChunkRows;
#pragma unroll
        for (int j = 0; j < kChunkRows / 8 / kWgWaves; ++j) {
            const int piece = wave * (kChunkRows / 8 / kWgWaves) + j;
            stage_piece(t, base + piece * 1024, piece * 8, rbase + piece * 8, row_end, lane);
        }
    };
    auto stage_c = [&](int chunk, int buf) {
        if (tid >= kChunkRows) return;
        const int gr = row0 + chunk * kChunkRows + tid;
        const int tn = tnorm[min(gr, row_end - 1)];
        const int tile16 = ((gr - row0) >> 5) & (kSegTiles - 1);
        const int i = tid & 31, slot = (i & 3) | ((i >> 3) << 2);
        const int c = gr < row_end ? (-tn) * 256 + ((15 - tile16) << 4) + (15 - slot) : INT_MIN;
        ((int*)(smem + buf * kBufBytes + kChunkRows * kDim))[tid] = c;
    };

    Top2 run[kSets];
#pragma unroll
    for (int set = 0; set < kSets; ++set) { run[set].k1 = run[set].k2 = INT_MIN; run[set].j1 = run[set].j2 = -1; }
    int m1[kSets][kChains], m2[kSets][kChains];  // [set][chain] packed running top-2 of the current segment
#pragma unroll
    for (int set = 0; set < kSets; ++set)
#pragma unroll
        for (int c = 0; c < kChains; ++c) m1[set][c] = m2[set][c] = INT_MIN;

    auto fold = [&](int seg_row_base) {
#pragma unroll
        for (int set = 0; set < kSets; ++set) {
            int n1 = m1[set][0], n2 = m2[set][0];
#pragma unroll
            for (int c = 1; c < kChains; ++c) {
                const int b1 = m1[set][c], b2 = m2[set][c];
                n2 = max3i(min(n1, b1), n2, b2);
                n1 = max(n1, b1);
            }
            const int pv[2] = {n1, n2};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = pv[e];
                const int slot = 15 - (p & 15), tile16 = 15 - ((p >> 4) & 15);
                const int i = (slot & 3) + ((slot >> 2) << 3) + 4 * h;
                const int j = p == INT_MIN ? -1 : seg_row_base + tile16 * kTileRows + i;
                insert_later(run[set], p >> 8, j);
            }
#pragma unroll
            for (int c = 0; c < kChains; ++c) m1[set][c] = m2[set][c] = INT_MIN;
        }
    };

    // top-2 update of one 32x32 tile of dot products, two keys at a time: with m1 >= m2 the two largest
    // of {m1, m2, p0, p1} are max3(m1, p0, p1) and max(m2, med3(m1, p0, p1)): 2.5 VALU ops per element
    // (key build included)
    auto epilogue = [&](const v16i& acc, const v4i* cv, int set) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int p0 = (acc[r] << 9) + cv[r >> 2][r & 3];
            const int p1 = (acc[r + 1] << 9) + cv[r >> 2][(r & 3) + 1];
            const int c = (r >> 1) & (kChains - 1);
            m2[set][c] = max(m2[set][c], med3i(m1[set][c], p0, p1));
            m1[set][c] = max3i_op(m1[set][c], p0, p1);
        }
    };

    if (nchunks > 0) {
        stage_rows(0, 0);
        stage_c(0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) stage_rows(chunk + 1, buf ^ 1);   // DMA runs under the sweep below

        const char* base = smem + buf * kBufBytes;
        const int* cbase = (const int*)(base + kChunkRows * kDim);
        const char* fbase = base + qrow * kDim;
        int foff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + h) ^ ((qrow >> 1) & 7)) << 4;

#if VO_KNN_PIPE
        // Software pipeline inside the wave: the four MFMAs of unit u+1 (a 32x32 tile of one query set)
        // are issued BEFORE the top-2 epilogue of unit u, whose VALU work then runs while the matrix
        // pipe is busy.  The register allocation leaves two waves per SIMD, both of this workgroup and
        // nearly in phase, so overlap across waves alone does not happen.
        {
            constexpr int kUnits = kChunkTiles * kSets;
            v4i va[4], cvb[2][4];
            v16i accb[2];
            auto load_tile = [&](int tile, int slot) {
#pragma unroll
                for (int s = 0; s < 4; ++s) va[s] = *(const v4i*)(fbase + tile * kTileRows * kDim + foff[s]);
#pragma unroll
                for (int g = 0; g < 4; ++g) cvb[slot][g] = *(const v4i*)(cbase + tile * kTileRows + 8 * g + 4 * h);
            };
            auto issue = [&](int set, int slot) {
                v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(va[s], qb[set][s], acc, 0, 0, 0);
                accb[slot] = acc;
            };
            load_tile(0, 0);
            issue(0, 0);
#pragma unroll
            for (int u = 0; u < kUnits; ++u) {
                const int tile = u / kSets, set = u % kSets;
                if (u + 1 < kUnits) {
                    const int ntile = (u + 1) / kSets, nset = (u + 1) % kSets;
                    if (nset == 0) load_tile(ntile, ntile & 1);
                    issue(nset, (u + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                epilogue(accb[u & 1], cvb[tile & 1], set);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        // no software pipelining: 4 waves per SIMD (two 8-wave workgroups per CU) cover the MFMA
        // chain latency and the LDS reads of one wave with the top-2 VALU work of the others
#pragma unroll
        for (int tile = 0; tile < kChunkTiles; ++tile) {
            v4i va[4], cv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) va[s] = *(const v4i*)(fbase + tile * kTileRows * kDim + foff[s]);
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[g] = *(const v4i*)(cbase + tile * kTileRows + 8 * g + 4 * h);
#pragma unroll
            for (int set = 0; set < kSets; ++set) {
                v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(va[s], qb[set][s], acc, 0, 0, 0);
                epilogue(acc, cv, set);
            }
        }

#endif
        if (chunk + 1 < nchunks) stage_c(chunk + 1, buf ^ 1);
        if (((chunk + 1) & (kSegTiles / kChunkTiles - 1)) == 0 || chunk + 1 == nchunks)
            fold(row0 + (chunk / (kSegTiles / kChunkTiles)) * (kSegTiles * kTileRows));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- the two lane halves hold the same query over interleaved train rows: merge
#pragma unroll
    for (int set = 0; set < kSets; ++set) {
        Top2 o;
        o.k1 = __shfl_xor(run[set].k1, 32); o.j1 = __shfl_xor(run[set].j1, 32);
        o.k2 = __shfl_xor(run[set].k2, 32); o.j2 = __shfl_xor(run[set].j2, 32);
        const Top2 mrg = merge2(run[set], o);
        const int qi = qbase + set * 32 + qrow;
        if (h == 0 && qi < nq) {
            if (nsplit == 1) {
                // one split sees every train row: finish here, no merge launch
                const int qn = qnorm[(size_t)b * nq + qi];
                const size_t o = ((size_t)b * nq + qi) * 2;
                *(int2*)(idx + o) = make_int2(mrg.j1, mrg.j2);
                *(int2*)(d2 + o) = make_int2(mrg.j1 >= 0 ? qn - mrg.k1 : INT_MAX, mrg.j2 >= 0 ? qn - mrg.k2 : INT_MAX);
            } else {
                partial[((size_t)b * nsplit + split) * nq + qi] = make_int4(mrg.k1, mrg.j1, mrg.k2, mrg.j2);
            }
        }
    }
}

__global__ __launch_bounds__(256)
void knn2_merge_kernel(const int4* __restrict__ partial, const int32_t* __restrict__ qnorm,
                       int nq, int nsplit, int32_t* __restrict__ idx, int32_t* __restrict__ d2)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (qi >= nq) return;
    Top2 r;
    r.k1 = r.k2 = INT_MIN; r.j1 = r.j2 = -1;
    for (int s = 0; s < nsplit; ++s) {
        const int4 p = partial[((size_t)b * nsplit + s) * nq + qi];
        insert_later(r, p.x, p.y);
        insert_later(r, p.z, p.w);
    }
    const int qn = qnorm[(size_t)b * nq + qi];
    const size_t o = ((size_t)b * nq + qi) * 2;
    *(int2*)(idx + o) = make_int2(r.j1, r.j2);
    *(int2*)(d2 + o) = make_int2(r.j1 >= 0 ? qn - r.k1 : INT_MAX, r.j2 >= 0 ? qn - r.k2 : INT_MAX);
}

// f32 (integers 0..255) -> int8 (value-128) + squared norm of the int8 row; 32 lanes per row, four rows
// per lane group with their loads issued together (the kernel is a pure HBM stream: 512 B in, 132 B out
// per row, and one 16-byte load per thread in flight does not cover the latency)
constexpr int kDescRowsPerGroup = 4;
__global__ __launch_bounds__(256)
void desc_f32_to_i8_kernel(const float* __restrict__ desc, int n, int8_t* __restrict__ out,
                           int32_t* __restrict__ norm, int32_t* __restrict__ flags)
{
    const int row0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * kDescRowsPerGroup, l = threadIdx.x & 31;
    float4 v[kDescRowsPerGroup];
#pragma unroll
    for (int r = 0; r < kDescRowsPerGroup; ++r) {
        const int row = row0 + r < n ? row0 + r : n - 1;
        v[r] = *(const float4*)(desc + (size_t)row * kDim + 4 * l);
    }
    int bad = 0;
#pragma unroll
    for (int r = 0; r < kDescRowsPerGroup; ++r) {
        const int row = row0 + r;
        const float f[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
        int s = 0, su = 0;
        signed char ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = f[k];
            if (row < n && (!(x >= 0.f && x <= 255.f) || x != truncf(x))) bad = VO_DESC_BAD_VALUE;
            const int u = (int)fminf(fmaxf(x, 0.f), 255.f);
            const int c = u - 128;
            ov[k] = (signed char)c;
            s += c * c;
            su += u * u;
        }
        if (row < n) {
            char4 o;
            o.x = ov[0]; o.y = ov[1]; o.z = ov[2]; o.w = ov[3];
            *(char4*)(out + (size_t)row * kDim + 4 * l) = o;
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            s += __shfl_xor(s, off);
            su += __shfl_xor(su, off);
        }
        if (row < n && l == 0) {
            norm[row] = s;
            if (su > (1 << 20)) bad |= VO_DESC_BAD_NORM;
        }
    }
    if (bad) atomicOr(flags, bad);
}

// grid-stride, one atomic per workgroup: a counter bumped once per wave serialises the whole launch
// on one L2 channel (~10 ns per atomic)
__global__ __launch_bounds__(256)
void ratio_test_kernel(const int32_t* __restrict__ d2, int n, double ratio,
                       uint8_t* __restrict__ keep, int32_t* __restrict__ count)
{
    __shared__ int s_cnt[4];
    int mine = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 d = *(const int2*)(d2 + 2 * (size_t)i);
        int k = 0;
        if (d.x != INT_MAX && d.y != INT_MAX) {
            const float m = __fsqrt_rn((float)d.x), s = __fsqrt_rn((float)d.y);
            k = (double)m < ratio * (double)s;
        }
        keep[i] = (uint8_t)k;
        mine += k;
    }
    for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (t) atomicAdd(count, t);
    }
}

struct KnnPlan { int nsplit, rows_per_split, nqb; };

int cu_count()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
}

KnnPlan knn_plan(int nq, int nt, int batch)
{
    KnnPlan p;
    p.nqb = vo_ceil_div(nq > 0 ? nq : 1, kWgQ);
    static const long per_cu = getenv("VO_KNN_WG_PER_CU") ? atol(getenv("VO_KNN_WG_PER_CU")) : 2;   // tuning knob
    const long want = per_cu * cu_count();                   // two 8-wave workgroups per CU
    long ns = (want + (long)p.nqb * batch - 1) / ((long)p.nqb * batch);
    const int max_ns = vo_ceil_div(nt > 0 ? nt : 1, kSegTiles * kTileRows);
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    int rps = vo_ceil_div(nt > 0 ? nt : 1, (int)ns);
    rps = vo_ceil_div(rps, kChunkRows) * kChunkRows;
    p.rows_per_split = rps;
    p.nsplit = vo_ceil_div(nt > 0 ? nt : 1, rps);
    return p;
}

}  // namespace

extern "C" size_t vo_knn2_workspace_bytes(int nq, int nt, int batch)
{
    if (nq <= 0 || nt <= 0 || batch <= 0) return 16;
    const KnnPlan p = knn_plan(nq, nt, batch);
    return (size_t)batch * p.nsplit * nq * sizeof(int4);
}

extern "C" int vo_knn2_l2_i8(const int8_t* q, const int32_t* qnorm, int nq,
                             const int8_t* t, const int32_t* tnorm, int nt,
                             int32_t* idx, int32_t* d2, int batch,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    VO_CHECK_ARG(nq >= 0 && nt >= 0 && batch >= 0, VO_EINVAL, "vo_knn2_l2_i8: negative size");
    if (nq == 0 || batch == 0) return VO_OK;
    VO_CHECK_ARG(q && qnorm && idx && d2, VO_EINVAL, "vo_knn2_l2_i8: null query/output pointer");
    VO_CHECK_ARG(nt <= VO_KNN_MAX_TRAIN, VO_ESIZE, "vo_knn2_l2_i8: nt=%d exceeds %d train rows", nt, VO_KNN_MAX_TRAIN);
    VO_CHECK_ARG(nt == 0 || (t && tnorm), VO_EINVAL, "vo_knn2_l2_i8: null train pointer");
    VO_CHECK_ARG(batch <= 65535, VO_ESIZE, "vo_knn2_l2_i8: batch=%d exceeds 65535", batch);
    VO_CHECK_ARG((((uintptr_t)q | (uintptr_t)t) & 15) == 0 && (((uintptr_t)idx | (uintptr_t)d2) & 7) == 0,
                 VO_EINVAL, "vo_knn2_l2_i8: descriptors must be 16-byte aligned, outputs 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const KnnPlan p = knn_plan(nq, nt, batch);
    const size_t need = vo_knn2_workspace_bytes(nq, nt, batch);
    VO_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, VO_EWORKSPACE,
                 "vo_knn2_l2_i8: workspace of %zu bytes (16-byte aligned) required, got %zu", need, workspace_bytes);
    VO_CHECK_ARG(p.nsplit <= 65535, VO_ESIZE, "vo_knn2_l2_i8: too many train splits");
    int4* partial = (int4*)workspace;
    if (kLdsBytes > 65536) {                         // large dynamic LDS needs an opt-in
        static bool lds_attr_set = false;
        if (!lds_attr_set) {
            VO_HIP_TRY(hipFuncSetAttribute((const void*)knn2_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
            lds_attr_set = true;
        }
    }
    dim3 grid(p.nqb, p.nsplit, batch);
    vo_prof_begin(VO_STAGE_KNN, 0x1000, s);
    hipLaunchKernelGGL(knn2_partial_kernel, grid, dim3(kWgThreads), kLdsBytes, s, q, t, tnorm, nq, nt,
                       p.rows_per_split, p.nsplit, partial, qnorm, idx, d2);
    vo_prof_end(VO_STAGE_KNN, s);
    VO_LAUNCH_CHECK();
    if (p.nsplit == 1) return VO_OK;
    hipLaunchKernelGGL(knn2_merge_kernel, dim3(vo_ceil_div(nq, 256), batch), dim3(256), 0, s, partial, qnorm,
                       nq, p.nsplit, idx, d2);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_desc_f32_to_i8(const float* desc, int n, int8_t* out_i8, int32_t* out_norm,
                                 int32_t* flags, void* stream)
{
    VO_CHECK_ARG(n >= 0, VO_EINVAL, "vo_desc_f32_to_i8: negative row count");
    if (n == 0) return VO_OK;
    VO_CHECK_ARG(desc && out_i8 && out_norm && flags, VO_EINVAL, "vo_desc_f32_to_i8: null pointer");
    VO_CHECK_ARG(((uintptr_t)desc & 15) == 0 && ((uintptr_t)out_i8 & 3) == 0, VO_EINVAL,
                 "vo_desc_f32_to_i8: desc must be 16-byte aligned");
    hipLaunchKernelGGL(desc_f32_to_i8_kernel, dim3(vo_ceil_div(n, 8 * kDescRowsPerGroup)), dim3(256), 0, (hipStream_t)stream,
                       desc, n, out_i8, out_norm, flags);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_ratio_test(const int32_t* d2, int n, double ratio, uint8_t* keep, int32_t* count, void* stream)
{
    VO_CHECK_ARG(n >= 0, VO_EINVAL, "vo_ratio_test: negative size");
    if (n == 0) return VO_OK;
    VO_CHECK_ARG(d2 && keep && count, VO_EINVAL, "vo_ratio_test: null pointer");
    const int blocks = vo_ceil_div(n, 256) < 1024 ? vo_ceil_div(n, 256) : 1024;
    hipLaunchKernelGGL(ratio_test_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       d2, n, ratio, keep, count);
    VO_LAUNCH_CHECK();
    return VO_OK;
}
