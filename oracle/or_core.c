/*
 * or_core.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * Restates, from SURVEY.md Appendix A.1/A.2 (OpenCV 4.6.0 semantics, OpenCV itself absent
 * => parity unpinned):
 *   - BFMatcher(NORM_L2).knnMatch(k=2)          (reference call site src/features.py:100-101)
 *   - cv::RNG, PointSetRegistrator getSubset, RANSACUpdateNumIters, findInliers and the
 *     best-model bookkeeping of RANSACPointSetRegistrator::run
 *     (reference call sites src/utils/geometry.py:39,44 and src/pnp.py:16-26)
 */
#include "vo_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_det_math = 0;
void vo_or_set_det_math(int on) { g_det_math = on ? 1 : 0; }
int vo_or_get_det_math(void) { return g_det_math; }

/* ------------------------------------------------------------------ A.1 2-NN */
/* f32 accumulation in 8 interleaved partial sums (the shape of an 8-lane SIMD loop, as
 * OpenCV's batchDistance uses); exact for SIFT's integer-valued data whatever the order
 * because every partial sum stays below 2^24. */
static float l2sqr_f32(const float* a, const float* b, int dim)
{
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    for (; c + 8 <= dim; c += 8)
        for (int l = 0; l < 8; ++l) {
            float d = a[c + l] - b[c + l];
            s[l] += d * d;
        }
    float r = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; c < dim; ++c) {
        float d = a[c] - b[c];
        r += d * d;
    }
    return r;
}

int vo_or_knn2(const float* q, int nq, const float* t, int nt, int dim,
               int32_t* idx, float* dist)
{
    if (nq < 0 || nt < 0 || dim <= 0) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int32_t i0 = -1, i1 = -1;
        const float* qi = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            float d = sqrtf(l2sqr_f32(qi, t + (size_t)j * dim, dim));
            if (d < d1) {            /* strict: an equal candidate never displaces */
                if (d0 > d) {        /* strict: equal keeps the lower train index first */
                    d1 = d0; i1 = i0;
                    d0 = d;  i0 = j;
                } else {
                    d1 = d;  i1 = j;
                }
            }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
    return 0;
}

/* ------------------------------------------------------------------ A.2 cv::RNG */
typedef struct { uint64_t state; } or_rng;

static void rng_init(or_rng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffULL; }
static uint32_t rng_next(or_rng* r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
static int rng_uniform(or_rng* r, int a, int b)
{
    return a == b ? a : (int)(rng_next(r) % (uint32_t)(b - a) + (uint32_t)a);
}

void vo_or_rng_stream(uint64_t seed, int n, uint32_t* out)
{
    or_rng r; rng_init(&r, seed);
    for (int i = 0; i < n; ++i) out[i] = rng_next(&r);
}

/* haveCollinearPoints: the last of `count` points against every earlier pair. The
 * coordinate differences are f32 subtractions promoted to f64. */
static int have_collinear(const float* p, int count)
{
    int i = count - 1;
    for (int j = 0; j < i; ++j) {
        double dx1 = (double)(p[2 * j] - p[2 * i]);
        double dy1 = (double)(p[2 * j + 1] - p[2 * i + 1]);
        for (int k = 0; k < j; ++k) {
            double dx2 = (double)(p[2 * k] - p[2 * i]);
            double dy2 = (double)(p[2 * k + 1] - p[2 * i + 1]);
            if (fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

int vo_or_fund_check_subset(const float* ms1, const float* ms2, int count)
{
    return !have_collinear(ms1, count) && !have_collinear(ms2, count);
}

/* one getSubset(..., maxAttempts) call; returns 1 if a subset was accepted.  RANSACPointSetRegistrator::run passes
 * 10000, LMeDSPointSetRegistrator::run leaves the declaration's default of 1000 (ptsetreg.cpp, 4.6.0) */
static int get_subset(or_rng* rng, int count, int mp, int check_mode,
                      const float* m1, const float* m2, int32_t* idx, int max_attempts)
{
    float s1[2 * 16], s2[2 * 16];
    for (int attempt = 0; attempt < max_attempts; ++attempt) {
        for (int i = 0; i < mp; ++i) {
            int v;
            for (;;) {
                v = rng_uniform(rng, 0, count);
                int dup = 0;
                for (int k = 0; k < i; ++k) if (idx[k] == v) { dup = 1; break; }
                if (!dup) break;
            }
            idx[i] = v;
            if (check_mode == 1) {
                s1[2 * i] = m1[2 * v]; s1[2 * i + 1] = m1[2 * v + 1];
                s2[2 * i] = m2[2 * v]; s2[2 * i + 1] = m2[2 * v + 1];
            }
        }
        if (check_mode != 1 || vo_or_fund_check_subset(s1, s2, mp)) return 1;
    }
    return 0;
}

int vo_or_gen_subsets(int count, int model_points, int max_iters, int check_mode,
                      const float* m1, const float* m2, int32_t* subsets)
{
    return vo_or_gen_subsets_attempts(count, model_points, max_iters, check_mode, m1, m2, subsets, 10000);
}

int vo_or_gen_subsets_attempts(int count, int model_points, int max_iters, int check_mode,
                               const float* m1, const float* m2, int32_t* subsets, int max_attempts)
{
    if (count < model_points || model_points > 16) return 0;
    or_rng rng; rng_init(&rng, (uint64_t)-1);
    int niters = max_iters > 1 ? max_iters : 1;
    int it = 0;
    for (; it < niters; ++it)
        if (!get_subset(&rng, count, model_points, check_mode, m1, m2,
                        subsets + (size_t)it * model_points, max_attempts))
            break;
    return it;
}

static int cv_round(double v) { return (int)lrint(v); }   /* round-half-even */

int vo_or_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.;   p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : cv_round(num / denom);
}

void vo_or_ransac_replay(const int32_t* counts, const int32_t* nmodels, int n_gen,
                         int max_models, int count, int model_points,
                         double confidence, int max_iters, int32_t* out)
{
    int niters = max_iters > 1 ? max_iters : 1;
    int max_good = 0, best_iter = -1, best_k = -1;
    for (int iter = 0; iter < niters && iter < n_gen; ++iter) {
        int nm = nmodels[iter];
        for (int k = 0; k < nm; ++k) {
            int good = counts[(size_t)iter * max_models + k];
            int floor_ = max_good > model_points - 1 ? max_good : model_points - 1;
            if (good > floor_) {
                max_good = good; best_iter = iter; best_k = k;
                niters = vo_or_update_num_iters(confidence, (double)(count - good) / count,
                                                model_points, niters);
            }
        }
    }
    out[0] = best_iter; out[1] = best_k; out[2] = max_good; out[3] = niters;
}

int vo_or_find_inliers(const float* err, int n, double thresh, uint8_t* mask)
{
    float t = (float)(thresh * thresh);
    int good = 0;
    for (int i = 0; i < n; ++i) {
        int f = err[i] <= t;
        if (mask) mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}
