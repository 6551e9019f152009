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

// one wave; counts laid out [iter][model]; scans in (iter, model) order
__global__ __launch_bounds__(64)
void ransac_replay_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ nmodels,
                          int n_iters, int max_models, int n_points, int model_points,
                          double confidence, int max_iters, int32_t* __restrict__ result)
{
    const int lane = threadIdx.x;
    int niters = max_iters > 1 ? max_iters : 1;
    int max_good = 0, best_iter = -1, best_k = -1;
    long pos = 0;                                         // flat index iter*max_models + k
    for (;;) {
        // "iter < niters" is only tested between iterations: the models of the iteration
        // that produced the latest improvement are all scored even if niters just shrank
        int lim_it = niters < n_iters ? niters : n_iters;
        if (best_iter + 1 > lim_it) lim_it = best_iter + 1;
        const long limit = (long)lim_it * max_models;
        if (pos >= limit) break;
        const long i = pos + lane;
        const int floor_ = max_good > model_points - 1 ? max_good : model_points - 1;
        bool hit = false;
        int c = 0;
        if (i < limit) {
            const int it = (int)(i / max_models), k = (int)(i - (long)it * max_models);
            c = counts[i];
            const bool present = nmodels ? k < nmodels[it] : c >= 0;
            hit = present && c > floor_;
        }
        const unsigned long long bal = __ballot(hit);
        if (bal == 0) { pos += 64; continue; }
        const int first = __ffsll((long long)bal) - 1;
        const int good = __shfl(c, first);
        const long at = pos + first;
        max_good = good;
        best_iter = (int)(at / max_models);
        best_k = (int)(at - (long)best_iter * max_models);
        niters = update_num_iters(confidence, (double)(n_points - good) / n_points, model_points, niters);
        pos = at + 1;
    }
    if (lane == 0) {
        result[0] = best_iter; result[1] = best_k; result[2] = max_good; result[3] = niters;
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

extern "C" int vo_ransac_replay(const int32_t* counts, const int32_t* nmodels, int n_iters, int max_models,
                                int n_points, int model_points, double confidence, int max_iters,
                                int32_t* result, void* stream)
{
    VO_CHECK_ARG(n_iters >= 0 && max_models >= 1 && n_points >= 1 && model_points >= 1, VO_EINVAL,
                 "vo_ransac_replay: bad sizes");
    VO_CHECK_ARG(result && (counts || n_iters == 0), VO_EINVAL, "vo_ransac_replay: null pointer");
    VO_CHECK_ARG(confidence > 0 && confidence < 1, VO_EINVAL, "vo_ransac_replay: confidence must be in (0,1)");
    hipLaunchKernelGGL(ransac_replay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counts, nmodels, n_iters,
                       max_models, n_points, model_points, confidence, max_iters, result);
    VO_LAUNCH_CHECK();
    return VO_OK;
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
    VO_CHECK_ARG(model_points >= 1 && model_points <= 16, VO_EINVAL, "vo_host_gen_subsets: model_points out of range");
    VO_CHECK_ARG(subsets_host != nullptr, VO_EINVAL, "vo_host_gen_subsets: null output");
    if (n_points < model_points) return 0;
    CvRng rng((uint64_t)-1);
    const int niters = max_iters > 1 ? max_iters : 1;
    int it = 0;
    for (; it < niters; ++it) {
        int32_t* idx = subsets_host + (size_t)it * model_points;
        bool found = false;
        for (int attempt = 0; attempt < 10000 && !found; ++attempt) {
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
