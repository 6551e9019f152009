// vo_ransac.hip -- the bookkeeping around scoring in OpenCV's RANSACPointSetRegistrator::run
// (calib3d ptsetreg.cpp), split so that the data-parallel part can
// run as one batch:
//   * host: the sample stream.  getSubset draws from cv::RNG((uint64)-1) and its draws never
//     depend on scores, so all max_iters subsets are generated up front (vo_host_gen_subsets);
//   * device: the sequential "first strict improvement, then shrink niters" scan over the
//     batch-computed inlier counts (vo_ransac_replay), one wave, 64 counts per step.
// Reference call sites: src/utils/geometry.py:39,44 and src/pnp.py:16-26.
#include "vo_common.h"
#include <float.h>
#include <math.h>
#include <mutex>

namespace {

// RANSACUpdateNumIters.  pow/log are the device library's; a result differs from glibc's only
// if num/denom lands within an ulp of a rounding boundary of cvRound.
__host__ __device__ inline int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
    return (int)rint(num / denom);          // cvRound: round half to even
}

// One workgroup (16 waves) per problem; counts laid out [iter][model] and visited in that order,
// 1024 entries per step.  An entry is a "record" when it beats everything before it (and
// modelPoints-1): exactly the entries the sequential loop would accept, as long as the loop is still
// running.  Records of a step are found with a wave prefix-max + ballot and handed, in order, to
// thread 0, which applies RANSACUpdateNumIters and stops the scan where OpenCV's loop would have.
// Two workgroup sizes: 1024 threads for long tables (fewest barrier rounds), 256 for short ones -- a
// 1024-thread workgroup cannot get a CU while another stream's big launch keeps refilling every slot
// with its own small workgroups, a 256-thread one can (the PnP chain runs beside the F scoring launch).
template <int kReplayThreads>
__global__ __launch_bounds__(kReplayThreads)
void ransac_replay_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ nmodels,
                          int n_iters, int max_models, int n_points, const int32_t* __restrict__ n_dev,
                          int model_points, double confidence, int max_iters, int32_t* __restrict__ result, int scan_iters)
{
    constexpr int kReplayWaves = kReplayThreads / 64;
    __shared__ int s_wmax[kReplayWaves], s_wcnt[kReplayWaves];
    __shared__ int2 s_rec[kReplayThreads];
    __shared__ int s_niters, s_good, s_best_iter, s_best_k, s_stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const long stride = (long)n_iters * max_models;
    // only the first scan_iters iterations can hold models yet (a block-wise run replays after every block)
    const long len = (long)(scan_iters < n_iters ? scan_iters : n_iters) * max_models;
    counts += (size_t)b * stride;
    if (nmodels) nmodels += (size_t)b * n_iters;
    const int npts = n_dev ? n_dev[b] : n_points;
    if (tid == 0) {
        s_niters = max_iters > 1 ? max_iters : 1;
        s_good = 0; s_best_iter = -1; s_best_k = -1;
        s_stop = npts <= 0;
    }
    __syncthreads();
    int carry = model_points - 1;                       // max over everything already visited
    for (long base = 0; base < len; base += kReplayThreads) {
        // "iter < niters" is tested between iterations only: the iteration that produced the
        // latest improvement is always finished
        const int it_first = (int)(base / max_models);
        const int lim = s_niters < n_iters ? s_niters : n_iters;
        if (s_stop || (it_first >= lim && it_first > s_best_iter)) break;
        const long i = base + tid;
        int v = INT32_MIN;
        if (i < len) {
            const int it = (int)(i / max_models), k = (int)(i - (long)it * max_models);
            const int c = counts[i];
            if (nmodels ? k < nmodels[it] : c >= 0) v = c;
        }
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x = x > y ? x : y;
        }
        if (lane == 63) s_wmax[wave] = x;
        const int prev = __shfl_up(x, 1);
        __syncthreads();
        int before = carry, all = carry;
#pragma unroll
        for (int w = 0; w < kReplayWaves; ++w) {
            const int m = s_wmax[w];
            if (w < wave) before = before > m ? before : m;
            all = all > m ? all : m;
        }
        if (lane > 0) before = before > prev ? before : prev;
        const bool rec = v > before;
        const unsigned long long bal = __ballot(rec);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = __popcll(bal & ((1ull << lane) - 1)), total = 0;
#pragma unroll
        for (int w = 0; w < kReplayWaves; ++w) {
            const int c = s_wcnt[w];
            if (w < wave) off += c;
            total += c;
        }
        if (rec) s_rec[off] = make_int2((int)(i - base), v);
        __syncthreads();
        if (tid == 0 && total > 0) {
            int niters = s_niters, good = s_good, best_iter = s_best_iter, best_k = s_best_k, stop = 0;
            for (int r = 0; r < total; ++r) {
                const long pos = base + s_rec[r].x;
                const int it = (int)(pos / max_models), k = (int)(pos - (long)it * max_models);
                const int l = niters < n_iters ? niters : n_iters;
                if (!(it < l || it == best_iter)) { stop = 1; break; }
                good = s_rec[r].y; best_iter = it; best_k = k;
                niters = update_num_iters(confidence, (double)(npts - good) / npts, model_points, niters);
            }
            s_niters = niters; s_good = good; s_best_iter = best_iter; s_best_k = best_k; s_stop = stop;
        }
        __syncthreads();
        carry = all;
    }
    if (tid == 0) {
        int32_t* r = result + 4 * (size_t)b;
        r[0] = s_best_iter; r[1] = s_best_k; r[2] = s_good; r[3] = s_niters;
    }
}

__global__ __launch_bounds__(256)
void argmax_first_kernel(const int32_t* __restrict__ counts, int n, unsigned long long* __restrict__ packed)
{
    // key = count (biased to unsigned) in the high word, ~index in the low word: the maximum
    // is the largest count at the lowest index, independent of evaluation order
    unsigned long long best = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key =
            ((unsigned long long)((unsigned)counts[i] ^ 0x80000000u) << 32) | (unsigned)(~i);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(packed, best);
}

__global__ void argmax_unpack_kernel(const unsigned long long* __restrict__ packed, int32_t* __restrict__ result)
{
    const unsigned long long p = *packed;
    if (p == 0) { result[0] = INT32_MIN; result[1] = -1; return; }
    result[0] = (int)((unsigned)(p >> 32) ^ 0x80000000u);
    result[1] = (int)(~(unsigned)p);
}

// ---- host side: cv::RNG + getSubset ----------------------------------------------------------
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + (uint32_t)a); }
};

// FMEstimatorCallback::checkSubset -> haveCollinearPoints on the last point of each image
bool collinear_last(const float* pts4, const int32_t* idx, int count, int off)
{
    const int i = count - 1;
    const float xi = pts4[4 * idx[i] + off], yi = pts4[4 * idx[i] + off + 1];
    for (int j = 0; j < i; ++j) {
        const double dx1 = (double)(pts4[4 * idx[j] + off] - xi);
        const double dy1 = (double)(pts4[4 * idx[j] + off + 1] - yi);
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)(pts4[4 * idx[k] + off] - xi);
            const double dy2 = (double)(pts4[4 * idx[k] + off + 1] - yi);
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

}  // namespace

int vo_launch_replay(const int32_t* counts, const int32_t* nmodels, int n_iters, int max_models, int n_points,
                     const int32_t* n_dev, int model_points, double confidence, int max_iters, int batch,
                     int32_t* result, hipStream_t s, int scan_iters)
{
    if (batch == 0) return VO_OK;
    if (scan_iters <= 0 || scan_iters > n_iters) scan_iters = n_iters;
    // measured in the pair pipeline (12 288-entry F tables beside the PnP side stream): 256 threads 9.31 ms per step,
    // 1024 threads 9.41 -- the big workgroups wait for a whole CU
    if ((long)n_iters * max_models > 65536)
        hipLaunchKernelGGL(ransac_replay_kernel<1024>, dim3(batch), dim3(1024), 0, s, counts, nmodels, n_iters,
                           max_models, n_points, n_dev, model_points, confidence, max_iters, result, scan_iters);
    else
        hipLaunchKernelGGL(ransac_replay_kernel<256>, dim3(batch), dim3(256), 0, s, counts, nmodels, n_iters,
                           max_models, n_points, n_dev, model_points, confidence, max_iters, result, scan_iters);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

// ---- fused batched pipelines: solve -> score -> replay -> mask for `batch` problems of one shape --
namespace {
struct RansacWs {
    double* models; double* aux; int32_t* nmodels; int32_t* counts; int32_t* list; int32_t* nlist; int32_t* nstart;
    float* bounds;                           // batch x 4: the F count kernel's per-problem coordinate bounds
    int32_t* clist; int32_t* ncl;            // F only: contender list of a pruned scoring stage (vo_launch_score_fund_pruned)
    size_t zero_from, zero_bytes, bytes;     // [zero_from, zero_from + zero_bytes) is cleared per call
};
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
RansacWs carve(void* base, int width, int mw, int aux_w, int H, int batch)
{
    RansacWs w;
    size_t off = 0;
    char* p = (char*)base;
    const size_t slots = (size_t)batch * H * width;
    w.models = (double*)(p + off); off += align_up(slots * mw * sizeof(double));
    w.aux = (double*)(p + off); off += align_up(slots * aux_w * sizeof(double));
    w.list = (int32_t*)(p + off); off += align_up(slots * sizeof(int32_t));
    w.nstart = (int32_t*)(p + off); off += align_up((size_t)batch * sizeof(int32_t));
    w.bounds = (float*)(p + off); off += align_up((size_t)batch * 4 * sizeof(float));
    w.clist = w.ncl = nullptr;
    if (width == 3) {                                    // F
        w.clist = (int32_t*)(p + off); off += align_up(slots * sizeof(int32_t));
        w.ncl = (int32_t*)(p + off); off += align_up(((size_t)batch * 17 + 1) * sizeof(int32_t));
    }
    w.zero_from = off;
    w.nmodels = (int32_t*)(p + off); off += align_up((size_t)batch * H * sizeof(int32_t));
    w.counts = (int32_t*)(p + off); off += align_up(slots * sizeof(int32_t));
    w.nlist = (int32_t*)(p + off); off += align_up((size_t)batch * sizeof(int32_t));
    w.zero_bytes = off - w.zero_from;
    w.bytes = off;
    return w;
}
const int kWidth[3] = {3, 10, 1}, kMw[3] = {9, 9, 12}, kAux[3] = {0, 0, 3}, kModelPoints[3] = {7, 5, 4};
}  // namespace

extern "C" size_t vo_ransac_workspace_bytes(int kind, int H, int batch)
{
    if (kind < 0 || kind > 2 || H <= 0 || batch <= 0) return 256;
    return carve(nullptr, kWidth[kind], kMw[kind], kAux[kind], H, batch).bytes;
}

extern "C" int vo_ransac_batch(int kind, const int32_t* samples, int H, const void* pts, const float* img,
                               const double* K4_host, int N, const int32_t* n_dev, int batch, float thr2,
                               double confidence, int max_iters, int first_block, int32_t* result,
                               double* best_model, double* best_rvec, uint8_t* mask, void* workspace,
                               size_t workspace_bytes, void* stream)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2, VO_EINVAL, "vo_ransac_batch: kind must be 0 (F), 1 (E) or 2 (PnP)");
    VO_CHECK_ARG(H >= 0 && N >= 0 && batch >= 0 && first_block >= 0, VO_EINVAL, "vo_ransac_batch: negative size");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(H > 0 && N > 0, VO_EINVAL, "vo_ransac_batch: needs at least one sample and one correspondence");
    VO_CHECK_ARG(samples && pts && result && mask, VO_EINVAL, "vo_ransac_batch: null pointer");
    VO_CHECK_ARG(kind != VO_KIND_PNP || (img && K4_host && ((uintptr_t)img & 7) == 0), VO_EINVAL,
                 "vo_ransac_batch: PnP needs img (8-byte aligned) and K4_host");
    VO_CHECK_ARG(kind != VO_KIND_FUND || ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_ransac_batch: pts must be 16-byte aligned");
    VO_CHECK_ARG(confidence > 0 && confidence < 1, VO_EINVAL, "vo_ransac_batch: confidence must be in (0,1)");
    VO_CHECK_ARG((long)H * batch <= 2147483647L / 16 && batch <= 65535, VO_ESIZE, "vo_ransac_batch: H x batch too large");
    const int width = kWidth[kind], mw = kMw[kind], aux_w = kAux[kind];
    const RansacWs w = carve(workspace, width, mw, aux_w, H, batch);
    VO_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= w.bytes, VO_EWORKSPACE,
                 "vo_ransac_batch: workspace of %zu bytes (16-byte aligned) required, got %zu", w.bytes, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    // nmodels = 0 (iterations never solved hold no model), counts = 0, nlist = 0
    VO_HIP_TRY(hipMemsetAsync((char*)workspace + w.zero_from, 0, w.zero_bytes, s));
    const int Htot = H * batch;
    if (kind == VO_KIND_FUND) {
        const int rcb = vo_launch_fund_bounds(pts, n_dev, N, batch, w.bounds, s);
        if (rcb != VO_OK) return rcb;
    }
    // iteration blocks: [0, first_block), then blocks 4x as long as everything before, up to H.  After
    // each block the replay tells which problems OpenCV's loop would already have left (niters <= block
    // end); their later blocks cost nothing but the launch.  F: blocks 15x as long -- behind the first block only
    // upper bounds are computed for most hypotheses (a third of the cost), and every block costs a replay.
    int h0 = 0;
    for (int stage = 0; h0 < H; ++stage) {
        int h1 = H;
        const long grow = kind == VO_KIND_FUND ? 16 : 5;
        if (first_block > 0 && first_block < H) h1 = stage == 0 ? first_block : (int)((long)h0 * grow < H ? (long)h0 * grow : H);
        if (H - h1 < h1 / 4) h1 = H;                     // do not leave a sliver for a last stage
        const int hlen = h1 - h0;
        const int32_t* prev = stage == 0 ? nullptr : result;
        if (stage > 0) VO_HIP_TRY(hipMemcpyAsync(w.nstart, w.nlist, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        int rc;
        vo_prof_begin(VO_STAGE_SOLVE, kind | (stage << 8), s);
        if (kind == VO_KIND_FUND)
            rc = vo_launch_solve_7pt(samples, Htot, H, (const float*)pts, N, n_dev, w.models, w.nmodels, w.list, w.nlist,
                                     h0, hlen, prev, s);
        else if (kind == VO_KIND_ESS)
            rc = vo_launch_solve_5pt(samples, Htot, H, (const double*)pts, N, n_dev, w.models, w.nmodels, w.list, w.nlist,
                                     h0, hlen, prev, s);
        else
            rc = vo_launch_solve_p3p(samples, Htot, H, (const float*)pts, img, K4_host, N, n_dev, w.models, w.aux, w.nmodels,
                                     w.list, w.nlist, h0, hlen, prev, s);
        vo_prof_end(VO_STAGE_SOLVE, s);
        if (rc != VO_OK) return rc;
        vo_prof_begin(VO_STAGE_SCORE, kind | (stage << 8), s);
        // F, blocks behind the first: the replay of the earlier blocks has left every problem's best count in `result`, and
        // only a hypothesis that can beat it needs an exact count (vo_score.hip: upper bounds from the matrix-core tier)
        if (kind == VO_KIND_FUND && stage > 0)
            rc = vo_launch_score_fund_pruned(w.models, w.list, w.nstart, w.nlist, H * width, pts, n_dev, N, batch, thr2, w.counts,
                                             hlen * width, w.bounds, result, w.clist, w.ncl, s);
        else
            rc = vo_launch_score(kind, w.models, nullptr, w.list, stage == 0 ? nullptr : w.nstart, w.nlist, H * width, pts, img,
                                 n_dev, N, batch, K4_host, thr2, w.counts, hlen * width, s, kind == VO_KIND_FUND ? w.bounds : nullptr);
        vo_prof_end(VO_STAGE_SCORE, s);
        if (rc != VO_OK) return rc;
        vo_prof_begin(VO_STAGE_REPLAY, kind | (stage << 8), s);
        rc = vo_launch_replay(w.counts, w.nmodels, H, width, N, n_dev, kModelPoints[kind], confidence, max_iters, batch, result, s, h1);
        vo_prof_end(VO_STAGE_REPLAY, s);
        if (rc != VO_OK) return rc;
        h0 = h1;
    }
    return vo_launch_mask_best(kind, w.models, H * width, width, aux_w ? w.aux : nullptr, aux_w, best_rvec, result, pts, img,
                               n_dev, N, batch, K4_host, thr2, mask, best_model, s);
}

// ---- frame pairs with 7..14 matches inside a batch ---------------------------------------------------
namespace {
int lmeds_iterations(double confidence, int max_iters)
{
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    const int n = update_num_iters(confidence, 0.45, 7, max_iters);     // LMeDSPointSetRegistrator: outlier ratio 0.45 assumed
    return n > 3 ? n : 3;
}
struct SmallWs { int32_t* samples; int32_t* n_lmeds; int32_t* n_small; void* ransac; size_t bytes; };
SmallWs small_carve(void* base, int H, int batch)
{
    SmallWs w;
    char* p = (char*)base;
    size_t off = 0;
    w.samples = (int32_t*)(p + off); off += align_up((size_t)batch * H * 7 * sizeof(int32_t));
    w.n_lmeds = (int32_t*)(p + off); off += align_up((size_t)batch * sizeof(int32_t));
    w.n_small = (int32_t*)(p + off); off += align_up((size_t)batch * sizeof(int32_t));
    w.ransac = p + off; off += carve(nullptr, 3, 9, 0, H, batch).bytes;
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" int vo_fund_small_iterations(double confidence, int max_iters) { return lmeds_iterations(confidence, max_iters); }

extern "C" size_t vo_fund_small_workspace_bytes(double confidence, int max_iters, int batch)
{
    if (batch <= 0) return 256;
    return small_carve(nullptr, lmeds_iterations(confidence, max_iters), batch).bytes;
}

extern "C" int vo_fund_small_batch(const float* pts4, int N, const int32_t* n_match, int batch, double confidence, int max_iters,
                                   int32_t* result, double* F, uint8_t* mask, double* models7, int32_t* nmodels7,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    VO_CHECK_ARG(N > 0 && batch >= 0 && batch <= 65535, VO_EINVAL, "vo_fund_small_batch: bad sizes");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(pts4 && n_match && result && F && mask && models7 && nmodels7 && ((uintptr_t)pts4 & 15) == 0, VO_EINVAL,
                 "vo_fund_small_batch: null/unaligned pointer");
    const int H = lmeds_iterations(confidence, max_iters);
    VO_CHECK_ARG((long)H * batch <= 2147483647L / 32, VO_ESIZE, "vo_fund_small_batch: iterations x batch too large");
    const SmallWs w = small_carve(workspace, H, batch);
    VO_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= w.bytes, VO_EWORKSPACE,
                 "vo_fund_small_batch: workspace of %zu bytes (16-byte aligned) required, got %zu", w.bytes, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const RansacWs r = carve(w.ransac, 3, 9, 0, H, batch);
    VO_HIP_TRY(hipMemsetAsync((char*)w.ransac + r.zero_from, 0, r.zero_bytes, s));
    VO_HIP_TRY(hipMemsetAsync(nmodels7, 0, (size_t)batch * sizeof(int32_t), s));
    int rc = vo_launch_small_prepare(n_match, batch, H, w.n_lmeds, w.n_small, nullptr, s);
    if (rc != VO_OK) return rc;
    // LMedS draws through getSubset's default of 1000 attempts (the RANSAC registrator passes 10000)
    rc = vo_launch_gen_subsets(w.n_lmeds, N, 7, H, pts4, batch, w.samples, nullptr, 1000, s);
    if (rc != VO_OK) return rc;
    rc = vo_launch_small_prepare(n_match, batch, H, nullptr, nullptr, w.samples, s);       // identity sample where n == 7
    if (rc != VO_OK) return rc;
    rc = vo_launch_solve_7pt(w.samples, H * batch, H, pts4, N, w.n_small, r.models, r.nmodels, r.list, r.nlist, 0, H, nullptr, s);
    if (rc != VO_OK) return rc;
    return vo_launch_small_finish(r.models, r.nmodels, H, pts4, N, w.n_small, batch, result, F, mask, models7, nmodels7, s);
}

// ---- hypothesis-sharded form (one RANSAC batch over several GPUs) -----------------
// Rank r solves and scores iterations [h_lo, h_hi) of every problem into the SAME workspace layout
// vo_ransac_batch uses; the ranks exchange their slices of the count tables (one all_gather on the
// host side, visual_odometry_amd/parallel.py), after which every rank holds the complete tables and
// vo_ransac_slice_finish replays them -- replicated, so every rank gets OpenCV's sequential answer.
extern "C" int vo_ransac_layout(int kind, int H, int batch, size_t* offsets_host)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2 && H > 0 && batch > 0 && offsets_host, VO_EINVAL, "vo_ransac_layout: bad arguments");
    const RansacWs w = carve(nullptr, kWidth[kind], kMw[kind], kAux[kind], H, batch);
    offsets_host[0] = (size_t)((char*)w.nmodels - (char*)nullptr);
    offsets_host[1] = (size_t)((char*)w.counts - (char*)nullptr);
    offsets_host[2] = (size_t)kWidth[kind];
    offsets_host[3] = w.bytes;
    return VO_OK;
}

extern "C" int vo_ransac_slice_counts(int kind, const int32_t* samples, int H, int h_lo, int h_hi, const void* pts,
                                      const float* img, const double* K4_host, int N, const int32_t* n_dev, int batch,
                                      float thr2, void* workspace, size_t workspace_bytes, void* stream)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2, VO_EINVAL, "vo_ransac_slice_counts: kind must be 0 (F), 1 (E) or 2 (PnP)");
    VO_CHECK_ARG(H > 0 && N > 0 && batch >= 0 && h_lo >= 0 && h_lo <= h_hi && h_hi <= H, VO_EINVAL,
                 "vo_ransac_slice_counts: bad sizes (H %d, slice [%d, %d), N %d, batch %d)", H, h_lo, h_hi, N, batch);
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(samples && pts, VO_EINVAL, "vo_ransac_slice_counts: null pointer");
    VO_CHECK_ARG(kind != VO_KIND_PNP || (img && K4_host && ((uintptr_t)img & 7) == 0), VO_EINVAL,
                 "vo_ransac_slice_counts: PnP needs img (8-byte aligned) and K4_host");
    VO_CHECK_ARG(kind != VO_KIND_FUND || ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_ransac_slice_counts: pts must be 16-byte aligned");
    VO_CHECK_ARG((long)H * batch <= 2147483647L / 16 && batch <= 65535, VO_ESIZE, "vo_ransac_slice_counts: H x batch too large");
    const int width = kWidth[kind], mw = kMw[kind], aux_w = kAux[kind];
    const RansacWs w = carve(workspace, width, mw, aux_w, H, batch);
    VO_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= w.bytes, VO_EWORKSPACE,
                 "vo_ransac_slice_counts: workspace of %zu bytes (16-byte aligned) required, got %zu", w.bytes, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    VO_HIP_TRY(hipMemsetAsync((char*)workspace + w.zero_from, 0, w.zero_bytes, s));
    const int hlen = h_hi - h_lo;
    if (hlen == 0) return VO_OK;
    int rc;
    if (kind == VO_KIND_FUND)
        rc = vo_launch_solve_7pt(samples, H * batch, H, (const float*)pts, N, n_dev, w.models, w.nmodels, w.list, w.nlist, h_lo,
                                 hlen, nullptr, s);
    else if (kind == VO_KIND_ESS)
        rc = vo_launch_solve_5pt(samples, H * batch, H, (const double*)pts, N, n_dev, w.models, w.nmodels, w.list, w.nlist, h_lo,
                                 hlen, nullptr, s);
    else
        rc = vo_launch_solve_p3p(samples, H * batch, H, (const float*)pts, img, K4_host, N, n_dev, w.models, w.aux, w.nmodels,
                                 w.list, w.nlist, h_lo, hlen, nullptr, s);
    if (rc != VO_OK) return rc;
    vo_prof_begin(VO_STAGE_SCORE, kind, s);
    rc = vo_launch_score(kind, w.models, nullptr, w.list, nullptr, w.nlist, H * width, pts, img, n_dev, N, batch, K4_host,
                         thr2, w.counts, hlen * width, s);
    vo_prof_end(VO_STAGE_SCORE, s);
    return rc;
}

extern "C" int vo_ransac_slice_finish(int kind, int H, int h_lo, int h_hi, int N, const int32_t* n_dev, int batch,
                                      double confidence, int max_iters, void* workspace, size_t workspace_bytes,
                                      int32_t* result, double* best_model, double* best_rvec, void* stream)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2, VO_EINVAL, "vo_ransac_slice_finish: kind must be 0 (F), 1 (E) or 2 (PnP)");
    VO_CHECK_ARG(H > 0 && N > 0 && batch >= 0 && h_lo >= 0 && h_lo <= h_hi && h_hi <= H, VO_EINVAL,
                 "vo_ransac_slice_finish: bad sizes");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(result && best_model, VO_EINVAL, "vo_ransac_slice_finish: null pointer");
    VO_CHECK_ARG(confidence > 0 && confidence < 1, VO_EINVAL, "vo_ransac_slice_finish: confidence must be in (0,1)");
    const int width = kWidth[kind], mw = kMw[kind], aux_w = kAux[kind];
    const RansacWs w = carve(workspace, width, mw, aux_w, H, batch);
    VO_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= w.bytes, VO_EWORKSPACE,
                 "vo_ransac_slice_finish: workspace of %zu bytes (16-byte aligned) required, got %zu", w.bytes, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int rc = vo_launch_replay(w.counts, w.nmodels, H, width, N, n_dev, kModelPoints[kind], confidence, max_iters, batch,
                                    result, s);
    if (rc != VO_OK) return rc;
    return vo_launch_select_owned(kind, w.models, H * width, width, aux_w ? w.aux : nullptr, aux_w, best_rvec, result, batch,
                                  h_lo, h_hi, best_model, s);
}

extern "C" int vo_ransac_mask_models(int kind, const double* best_model, const int32_t* result, const void* pts,
                                     const float* img, const double* K4_host, int N, const int32_t* n_dev, int batch,
                                     float thr2, uint8_t* mask, void* stream)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2 && N > 0 && batch >= 0, VO_EINVAL, "vo_ransac_mask_models: bad arguments");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(best_model && result && pts && mask, VO_EINVAL, "vo_ransac_mask_models: null pointer");
    VO_CHECK_ARG(kind != VO_KIND_PNP || (img && K4_host && ((uintptr_t)img & 7) == 0), VO_EINVAL,
                 "vo_ransac_mask_models: PnP needs img (8-byte aligned) and K4_host");
    VO_CHECK_ARG(kind != VO_KIND_FUND || ((uintptr_t)pts & 15) == 0, VO_EINVAL, "vo_ransac_mask_models: pts must be 16-byte aligned");
    return vo_launch_mask_direct(kind, best_model, result, pts, img, n_dev, N, batch, K4_host, thr2, mask, (hipStream_t)stream);
}

// ---- the allocation-free variant's workspace: one fixed array per GPU, in the code object -------
// `scratch` is a host-side stand-in here, not a device address: host code only ever hands it to
// hipGetSymbolAddress (once per device, with that device current) and uses the pointer that comes back.
#define VO_RANSAC_STATIC_BYTES ((size_t)VO_RANSAC_STATIC_MAX_BYTES)
static __device__ __align__(256) unsigned char scratch[VO_RANSAC_STATIC_MAX_BYTES];

namespace {
enum { kMaxDevices = 64 };
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices];            // device address of `scratch` on device d, or null

// address of the current device's copy of `scratch`; resolved on the first call per device
int scratch_of_current_device(void** out)
{
    int d = -1;
    VO_HIP_TRY(hipGetDevice(&d));
    VO_CHECK_ARG(d >= 0 && d < kMaxDevices, VO_ESIZE, "vo_ransac_batch_static: device ordinal %d not supported", d);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[d]) {
        void* p = nullptr;
        VO_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        VO_CHECK_ARG(p && ((uintptr_t)p & 255) == 0, VO_EHIP, "vo_ransac_batch_static: bad symbol address %p", p);
        g_scratch_ptr[d] = p;
    }
    *out = g_scratch_ptr[d];
    return VO_OK;
}
}  // namespace

extern "C" size_t vo_ransac_static_max_bytes(void) { return VO_RANSAC_STATIC_BYTES; }

extern "C" int vo_ransac_batch_static(int kind, const int32_t* samples, int H, const void* pts, const float* img,
                                      const double* K4_host, int N, const int32_t* n_dev, int batch, float thr2,
                                      double confidence, int max_iters, int first_block, int32_t* result,
                                      double* best_model, double* best_rvec, uint8_t* mask, void* stream)
{
    VO_CHECK_ARG(kind >= 0 && kind <= 2, VO_EINVAL, "vo_ransac_batch_static: kind must be 0 (F), 1 (E) or 2 (PnP)");
    VO_CHECK_ARG(H >= 0 && batch >= 0, VO_EINVAL, "vo_ransac_batch_static: negative size");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG((long)H * batch <= 2147483647L / 16, VO_ESIZE, "vo_ransac_batch_static: H x batch too large");
    const size_t need = H > 0 ? carve(nullptr, kWidth[kind], kMw[kind], kAux[kind], H, batch).bytes : 0;
    VO_CHECK_ARG(need <= VO_RANSAC_STATIC_BYTES, VO_ESIZE,
                 "vo_ransac_batch_static: kind %d with H = %d, batch = %d needs %zu bytes of scratch, the built-in array "
                 "holds %zu; use vo_ransac_batch with a caller-provided workspace", kind, H, batch, need,
                 VO_RANSAC_STATIC_BYTES);
    void* ws = nullptr;
    const int rc = scratch_of_current_device(&ws);
    if (rc != VO_OK) return rc;
    // the array keeps the previous call's contents; vo_ransac_batch clears what must start at zero
    return vo_ransac_batch(kind, samples, H, pts, img, K4_host, N, n_dev, batch, thr2, confidence, max_iters, first_block,
                           result, best_model, best_rvec, mask, ws, VO_RANSAC_STATIC_BYTES, stream);
}

extern "C" int vo_ransac_replay(const int32_t* counts, const int32_t* nmodels, int n_iters, int max_models,
                                int n_points, int model_points, double confidence, int max_iters,
                                int32_t* result, void* stream)
{
    VO_CHECK_ARG(n_iters >= 0 && max_models >= 1 && n_points >= 1 && model_points >= 1, VO_EINVAL,
                 "vo_ransac_replay: bad sizes");
    VO_CHECK_ARG(result && (counts || n_iters == 0), VO_EINVAL, "vo_ransac_replay: null pointer");
    VO_CHECK_ARG(confidence > 0 && confidence < 1, VO_EINVAL, "vo_ransac_replay: confidence must be in (0,1)");
    return vo_launch_replay(counts, nmodels, n_iters, max_models, n_points, nullptr, model_points, confidence, max_iters,
                            1, result, (hipStream_t)stream);
}

extern "C" int vo_argmax_first(const int32_t* counts, int n, int32_t* result, void* stream)
{
    VO_CHECK_ARG(n >= 0 && result, VO_EINVAL, "vo_argmax_first: bad arguments");
    VO_CHECK_ARG(((uintptr_t)result & 7) == 0, VO_EINVAL, "vo_argmax_first: result must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // result doubles as the 8-byte packed accumulator before it is unpacked in place
    VO_HIP_TRY(hipMemsetAsync(result, 0, 8, s));
    if (n > 0) {
        VO_CHECK_ARG(counts != nullptr, VO_EINVAL, "vo_argmax_first: null counts");
        int blocks = vo_ceil_div(n, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(argmax_first_kernel, dim3(blocks), dim3(256), 0, s, counts, n,
                           (unsigned long long*)result);
        VO_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(argmax_unpack_kernel, dim3(1), dim3(1), 0, s, (const unsigned long long*)result, result);
    VO_LAUNCH_CHECK();
    return VO_OK;
}

extern "C" int vo_host_gen_subsets(int n_points, int model_points, int max_iters, const float* pts4_host,
                                   int32_t* subsets_host)
{
    return vo_host_gen_subsets_attempts(n_points, model_points, max_iters, 10000, pts4_host, subsets_host);
}

extern "C" int vo_host_gen_subsets_attempts(int n_points, int model_points, int max_iters, int max_attempts,
                                            const float* pts4_host, int32_t* subsets_host)
{
    VO_CHECK_ARG(max_attempts >= 1, VO_EINVAL, "vo_host_gen_subsets_attempts: max_attempts must be positive");
    VO_CHECK_ARG(model_points >= 1 && model_points <= 16, VO_EINVAL, "vo_host_gen_subsets: model_points out of range");
    VO_CHECK_ARG(subsets_host != nullptr, VO_EINVAL, "vo_host_gen_subsets: null output");
    if (n_points < model_points) return 0;
    CvRng rng((uint64_t)-1);
    const int niters = max_iters > 1 ? max_iters : 1;
    int it = 0;
    for (; it < niters; ++it) {
        int32_t* idx = subsets_host + (size_t)it * model_points;
        bool found = false;
        for (int attempt = 0; attempt < max_attempts && !found; ++attempt) {
            for (int i = 0; i < model_points; ++i) {
                int v;
                for (;;) {
                    v = rng.uniform(0, n_points);
                    bool dup = false;
                    for (int k = 0; k < i; ++k) dup |= idx[k] == v;
                    if (!dup) break;
                }
                idx[i] = v;
            }
            found = !pts4_host || (!collinear_last(pts4_host, idx, model_points, 0) &&
                                   !collinear_last(pts4_host, idx, model_points, 2));
        }
        if (!found) break;
    }
    return it;
}

extern "C" int vo_host_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    return update_num_iters(p, ep, model_points, max_iters);
}
