// vo_common.h -- shared by the translation units of libvo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vo_hip.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

void vo_set_error(const char* fmt, ...);

#define VO_CHECK_ARG(cond, code, ...)                 \
    do {                                              \
        if (!(cond)) {                                \
            vo_set_error(__VA_ARGS__);                \
            return (code);                            \
        }                                             \
    } while (0)

#define VO_HIP_TRY(expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            vo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                         __FILE__, __LINE__);                                         \
            return VO_EHIP;                                                           \
        }                                                                             \
    } while (0)

#define VO_LAUNCH_CHECK() VO_HIP_TRY(hipGetLastError())

static inline int vo_ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- internal launchers shared between translation units (not part of the C ABI) ------------
enum { VO_KIND_FUND = 0, VO_KIND_ESS = 1, VO_KIND_PNP = 2 };
int vo_launch_init_counts(const uint8_t* valid, const int32_t* nmodels, int slots_per_sample, long total,
                          int32_t* counts, hipStream_t s);
int vo_launch_score(int kind, const double* models, const uint8_t* valid, const int32_t* list, const int32_t* nstart,
                    const int32_t* nlist, int H, const void* pts, const float* img, const int32_t* n_dev, int N, int batch,
                    const double* K4, float thr2, int32_t* counts, int max_slots, hipStream_t s,
                    const float* fund_bounds = nullptr);
// F scoring of list entries [nstart, nlist) behind a replayed block: upper bounds for everybody, exact counts for the
// models whose bound exceeds result[4 b + 2] (vo_score.hip); clist batch x Hs ints, ncl 17 batch + 1 ints of scratch
int vo_launch_score_fund_pruned(const double* models, const int32_t* list, const int32_t* nstart, const int32_t* nlist, int Hs,
                                const void* pts, const int32_t* n_dev, int N, int batch, float thr2, int32_t* counts,
                                int max_slots, const float* fund_bounds, const int32_t* result, int32_t* clist, int32_t* ncl,
                                hipStream_t s);
// per-problem bounds the F count kernel sets its models up with: (max |x1|,|y1|,1; max |x2|,|y2|,1; all coordinates tame ? 1 : 0; -)
int vo_launch_fund_bounds(const void* pts, const int32_t* n_dev, int N, int batch, float* bounds, hipStream_t s);
int vo_launch_mask_best(int kind, const double* models, int slots_per_problem, int slots_per_sample,
                        const double* aux, int aux_w, double* best_aux, const int32_t* result, const void* pts,
                        const float* img, const int32_t* n_dev, int N, int batch, const double* K4, float thr2,
                        uint8_t* mask, double* best_model, hipStream_t s);
// best_model[b] (and best_aux[b]) = the replay winner's model if its iteration lies in [own_lo, own_hi), else zeros
int vo_launch_select_owned(int kind, const double* models, int slots_per_problem, int slots_per_sample, const double* aux,
                           int aux_w, double* best_aux, const int32_t* result, int batch, int own_lo, int own_hi,
                           double* best_model, hipStream_t s);
// mask[b] from best_model[b] (batch x 9|12) for the problems with result[4b] >= 0, zeros otherwise
int vo_launch_mask_direct(int kind, const double* best_model, const int32_t* result, const void* pts, const float* img,
                          const int32_t* n_dev, int N, int batch, const double* K4, float thr2, uint8_t* mask, hipStream_t s);
int vo_launch_solve_7pt(const int32_t* samples, int H, int per, const float* pts, int N, const int32_t* n_dev,
                        double* models, int32_t* nmodels, int32_t* list, int32_t* nlist, int h0, int hlen,
                        const int32_t* prev, hipStream_t s);
int vo_launch_solve_5pt(const int32_t* samples, int H, int per, const double* pts, int N, const int32_t* n_dev,
                        double* models, int32_t* nmodels, int32_t* list, int32_t* nlist, int h0, int hlen,
                        const int32_t* prev, hipStream_t s);
int vo_launch_solve_p3p(const int32_t* samples, int H, int per, const float* obj, const float* img, const double* K4,
                        int N, const int32_t* n_dev, double* models, double* rvecs, int32_t* nmodels, int32_t* list,
                        int32_t* nlist, int h0, int hlen, const int32_t* prev, hipStream_t s);
int vo_launch_replay(const int32_t* counts, const int32_t* nmodels, int n_iters, int max_models, int n_points,
                     const int32_t* n_dev, int model_points, double confidence, int max_iters, int batch,
                     int32_t* result, hipStream_t s, int scan_iters = 0);

int vo_launch_gen_subsets(const int32_t* n_dev, int N, int model_points, int H, const float* pts4, int batch,
                          int32_t* samples, int32_t* n_generated, int max_attempts, hipStream_t s);
// frame pairs with 7..14 matches (vo_fund_small_batch): per-problem counts for the two regimes, identity samples for
// the 7-match case, and the LMedS ranking + result rows
int vo_launch_small_prepare(const int32_t* n_match, int batch, int H, int32_t* n_lmeds, int32_t* n_small, int32_t* samples,
                            hipStream_t s);
int vo_launch_small_finish(const double* models, const int32_t* nmodels, int H, const float* pts4, int N,
                           const int32_t* n_small, int batch, int32_t* result, double* F, uint8_t* mask, double* models7,
                           int32_t* nmodels7, hipStream_t s);
// timing aid (vo_runtime.hip): stage ids for vo_profile_enable's mask
enum { VO_STAGE_SOLVE = 0, VO_STAGE_SCORE = 1, VO_STAGE_REPLAY = 2, VO_STAGE_MASK = 3, VO_STAGE_KNN = 4 };
void vo_prof_begin(int stage, int tag, hipStream_t s);
void vo_prof_end(int stage, hipStream_t s);
