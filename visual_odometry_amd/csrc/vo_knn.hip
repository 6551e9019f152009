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
#define VO_KNN_TILES 6       /* 32-row train tiles per LDS stage */
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
constexpr int kSegChunks = 2;                                 // chunks per key segment (a power of two)
constexpr int kSegTiles = kSegChunks * kChunkTiles;           // tiles whose index fits the key's 4 tile bits
static_assert(kSegTiles <= 16, "tile index must fit 4 bits of the packed key");
constexpr int kSets = VO_KNN_SETS;                                      // 32-query sets per wave
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
    // the norm of row tid of a chunk is fetched at the START of the previous chunk's sweep (stage_c_load) and turned
    // into the key constant at its end (stage_c_store): the load's latency is not waited for
    auto stage_c_load = [&](int chunk) -> int {
        if (tid >= kChunkRows) return 0;
        const int gr = row0 + chunk * kChunkRows + tid;
        return tnorm[min(gr, row_end - 1)];
    };
    auto stage_c_store = [&](int chunk, int buf, int tn) {
        if (tid >= kChunkRows) return;
        asm volatile("" : "+v"(tn));                 // keeps the arithmetic (and with it the wait for the load) down here
        const int gr = row0 + chunk * kChunkRows + tid;
        const int tile16 = ((gr - row0) >> 5) % kSegTiles;
        const int i = tid & 31, slot = (i & 3) | ((i >> 3) << 2);
        const int c = gr < row_end ? (-tn) * 256 + ((15 - tile16) << 4) + (15 - slot) : INT_MIN;
        ((int*)(smem + buf * kBufBytes + kChunkRows * kDim))[tid] = c;
    };
    Top2 run[kSets];
#pragma unroll
    for (int set = 0; set < kSets; ++set) { run[set].k1 = run[set].k2 = INT_MIN; run[set].j1 = run[set].j2 = -1; }
    // packed running top-2 of the current key segment: two m1 chains per query set and their common m2
    int m1[kSets][2], m2[kSets];
#pragma unroll
    for (int set = 0; set < kSets; ++set) m1[set][0] = m1[set][1] = m2[set] = INT_MIN;

    auto fold = [&](int seg_row_base) {
#pragma unroll
        for (int set = 0; set < kSets; ++set) {
            // the two largest of the union: max of the chain maxima, then the larger of the other chain's maximum and m2
            const int n1 = max(m1[set][0], m1[set][1]);
            const int n2 = max(min(m1[set][0], m1[set][1]), m2[set]);
            const int pv[2] = {n1, n2};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = pv[e];
                const int slot = 15 - (p & 15), tile16 = 15 - ((p >> 4) & 15);
                const int i = (slot & 3) + ((slot >> 2) << 3) + 4 * h;
                const int j = p == INT_MIN ? -1 : seg_row_base + tile16 * kTileRows + i;
                insert_later(run[set], p >> 8, j);
            }
            m1[set][0] = m1[set][1] = m2[set] = INT_MIN;
        }
    };

    // Top-2 update of a quarter of one 32x32 tile of dot products (accumulator registers 4s .. 4s+3 = four train rows
    // of the lane's query), two keys at a time: with m1 >= m2 the two largest of {m1, m2, p0, p1} are
    // max3(m1, p0, p1) and max(m2, med3(m1, p0, p1)).  The two pairs of a quarter go to two independent m1 chains
    // (their instructions alternate, so no instruction waits for the one just issued) and share one m2 = the maximum
    // of every med3 so far: 4 key builds + 2 med3 + 2 max3 + 1 max3 = 9 VALU instructions per four rows.
    auto epilogue_part = [&](const v16i& acc, const v4i& c4, int set, int s) {
        const int p0 = (acc[4 * s] << 9) + c4[0], p1 = (acc[4 * s + 1] << 9) + c4[1];
        const int p2 = (acc[4 * s + 2] << 9) + c4[2], p3 = (acc[4 * s + 3] << 9) + c4[3];
        asm volatile("" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3));      // all four keys before the first selection
        const int t0 = med3i(m1[set][0], p0, p1);
        const int t1 = med3i(m1[set][1], p2, p3);
        m1[set][0] = max3i_op(m1[set][0], p0, p1);
        m1[set][1] = max3i_op(m1[set][1], p2, p3);
        m2[set] = max3i(m2[set], t0, t1);
    };

    if (nchunks > 0) {
        stage_rows(0, 0);
        stage_c_store(0, 0, stage_c_load(0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Software pipeline inside the wave, one MFMA at a time and one unit deep, running ACROSS the chunk barriers.
    // A unit is one (32-row tile, 32-query set) pair = four dependent MFMAs.  Block j of a chunk issues the four MFMAs
    // of unit j with a quarter of the top-2 epilogue of unit j-2 (8-9 VALU instructions) behind each of them: the VALU
    // work issues while the matrix pipe is busy (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles; issued as
    // a block of four, a wave sits in MFMA issue for ~100 cycles and only the last one overlaps anything), and the
    // accumulators it reads were finished a whole block ago, so no hazard wait states separate MFMA and reader (with
    // the epilogue only one MFMA behind its tile the compiler has to pad 40-cycle s_nops, measured: no gain at all).
    // Three accumulator sets rotate (unit j -> set j % 3; kUnits is a multiple of 3, so the rotation is static); the
    // last two units of a chunk are finished under the first two MFMA groups of the next chunk, after the barrier.
    // A fragments of tile k+1 and key constants of tile k are read from LDS while tile k is being multiplied.
    constexpr int kUnits = kChunkTiles * kSets;
    static_assert(kSets == 2 && kUnits % 3 == 0, "the accumulator rotation below assumes 2 query sets and 3 | units per chunk");
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v4i none4 = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
    v16i accb[3] = {zero16, zero16, zero16};
    v4i va[2][4], cv[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cv[(kChunkTiles - 1) & 1][g] = none4;     // "units -2, -1": keys INT_MIN, never selected

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        int tn_next = 0;
        if (chunk + 1 < nchunks) {
            stage_rows(chunk + 1, buf ^ 1);                        // DMA runs under the sweep below
            tn_next = stage_c_load(chunk + 1);
        }

        const char* base = smem + buf * kBufBytes;
        const int* cbase = (const int*)(base + kChunkRows * kDim);
        const char* fbase = base + qrow * kDim;
        int foff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) foff[s] = ((2 * s + h) ^ ((qrow >> 1) & 7)) << 4;
        auto load_va = [&](int tile) {
#pragma unroll
            for (int s = 0; s < 4; ++s) va[tile & 1][s] = *(const v4i*)(fbase + tile * kTileRows * kDim + foff[s]);
        };
        auto load_cv = [&](int tile) {
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[tile & 1][g] = *(const v4i*)(cbase + tile * kTileRows + 8 * g + 4 * h);
        };
        load_va(0);
        __builtin_amdgcn_sched_barrier(0);                 // tile 0's fragments first: the first MFMA waits for them only
#pragma unroll
        for (int j = 0; j < kUnits; ++j) {
            const int tile = j / kSets, set = j % kSets;
            // unit j-2: same query set, previous tile (for j = 0, 1 the last tile of the previous chunk)
            const int ptile = (tile + kChunkTiles - 1) % kChunkTiles;
            if (set == 0) {
                if (tile + 1 < kChunkTiles) load_va(tile + 1);
                load_cv(tile);                              // used from block j+2 on; its register set is free: cv(tile-2) is done
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                accb[j % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(va[tile & 1][s], qb[set][s], s == 0 ? zero16 : accb[j % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                epilogue_part(accb[(j + 1) % 3], cv[ptile & 1][s], set, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the previous chunk closed a key segment: its last two units have just been folded into m1/m2
            if (j == kSets - 1 && chunk > 0 && (chunk & (kSegChunks - 1)) == 0)
                fold(row0 + (chunk / kSegChunks - 1) * (kSegTiles * kTileRows));
        }

        if (chunk + 1 < nchunks) stage_c_store(chunk + 1, buf ^ 1, tn_next);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (nchunks > 0) {
        // drain: the last two units of the last chunk, then the open segment
#pragma unroll
        for (int j = kUnits; j < kUnits + kSets; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) epilogue_part(accb[(j + 1) % 3], cv[(kChunkTiles - 1) & 1][s], j % kSets, s);
        fold(row0 + ((nchunks - 1) / kSegChunks) * (kSegTiles * kTileRows));
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

// f32 (integers 0..255) -> int8 (value-128) + squared norm of the int8 row; 32 lanes per row, eight rows
// per lane group with their loads issued together (the kernel is a pure HBM stream: 512 B in, 132 B out
// per row, and one 16-byte load per thread in flight does not cover the latency); the loads are
// non-temporal -- every byte is read once.  Standalone, 2 M rows: 220 us = 6.1 TB/s (4 rows, plain loads: 233)
constexpr int kDescRowsPerGroup = 8;
__global__ __launch_bounds__(256)
void desc_f32_to_i8_kernel(const float* __restrict__ desc, int n, int8_t* __restrict__ out,
                           int32_t* __restrict__ norm, int32_t* __restrict__ flags)
{
    const int row0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * kDescRowsPerGroup, l = threadIdx.x & 31;
    float4 v[kDescRowsPerGroup];
#pragma unroll
    for (int r = 0; r < kDescRowsPerGroup; ++r) {
        const int row = row0 + r < n ? row0 + r : n - 1;
        {
            typedef float vf4 __attribute__((ext_vector_type(4)));
            const vf4 t = __builtin_nontemporal_load((const vf4*)(desc + (size_t)row * kDim + 4 * l));
            v[r] = make_float4(t.x, t.y, t.z, t.w);
        }
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
    // (a launch full of bad rows would otherwise serialise on this one word at ~10 ns per atomic: once the bits are set,
    // later lanes see them and skip the atomic; OR is idempotent, so the race is harmless)
    if (bad && (__atomic_load_n(flags, __ATOMIC_RELAXED) & bad) != bad) atomicOr(flags, bad);
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
