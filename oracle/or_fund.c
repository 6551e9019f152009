/*
 * or_fund.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * cv2.findFundamentalMat(p1, p2, cv2.RANSAC, thr, conf, maxIters) as called by the
 * reference at src/utils/geometry.py:39, restated from SURVEY.md Appendix A.2/A.3
 * (OpenCV 4.6.0 calib3d fundam.cpp / ptsetreg.cpp semantics; OpenCV absent => parity
 * unpinned): 7-point minimal solver, symmetric squared epipolar distance, sequential
 * RANSAC with adaptive iteration count, no final refit.
 */
#include "or_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void vo_or_fund_error(const double* F, const float* m1, const float* m2, int n, float* err)
{
    for (int i = 0; i < n; ++i) {
        double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        double a, b, c, d1, d2, s1, s2;
        a = F[0] * x1 + F[1] * y1 + F[2];
        b = F[3] * x1 + F[4] * y1 + F[5];
        c = F[6] * x1 + F[7] * y1 + F[8];
        s2 = 1. / (a * a + b * b);
        d2 = x2 * a + y2 * b + c;
        a = F[0] * x2 + F[3] * y2 + F[6];
        b = F[1] * x2 + F[4] * y2 + F[7];
        c = F[2] * x2 + F[5] * y2 + F[8];
        s1 = 1. / (a * a + b * b);
        d1 = x1 * a + y1 * b + c;
        double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
        err[i] = (float)(e1 < e2 ? e2 : e1);       /* std::max(e1, e2) */
    }
}

int vo_or_run7point(const float* ms1, const float* ms2, double* Fout, int hartley)
{
    double a[9 * 9], w[7], vt[7 * 7];
    double T1[3] = {1, 0, 0}, T2[3] = {1, 0, 0};   /* scale, cx, cy per image */
    memset(a, 0, sizeof(a));

    if (hartley) {
        double c1x = 0, c1y = 0, c2x = 0, c2y = 0, s1 = 0, s2 = 0;
        for (int i = 0; i < 7; ++i) {
            c1x += ms1[2 * i]; c1y += ms1[2 * i + 1];
            c2x += ms2[2 * i]; c2y += ms2[2 * i + 1];
        }
        c1x /= 7; c1y /= 7; c2x /= 7; c2y /= 7;
        for (int i = 0; i < 7; ++i) {
            double dx = ms1[2 * i] - c1x, dy = ms1[2 * i + 1] - c1y;
            s1 += sqrt(dx * dx + dy * dy);
            dx = ms2[2 * i] - c2x; dy = ms2[2 * i + 1] - c2y;
            s2 += sqrt(dx * dx + dy * dy);
        }
        if (s1 < FLT_EPSILON || s2 < FLT_EPSILON) return 0;
        T1[0] = sqrt(2.) / (s1 / 7); T1[1] = c1x; T1[2] = c1y;
        T2[0] = sqrt(2.) / (s2 / 7); T2[1] = c2x; T2[2] = c2y;
    }

    for (int i = 0; i < 7; ++i) {
        double x0 = ms1[2 * i], y0 = ms1[2 * i + 1], x1 = ms2[2 * i], y1 = ms2[2 * i + 1];
        if (hartley) {
            x0 = (x0 - T1[1]) * T1[0]; y0 = (y0 - T1[2]) * T1[0];
            x1 = (x1 - T2[1]) * T2[0]; y1 = (y1 - T2[2]) * T2[0];
        }
        double* r = a + i * 9;
        r[0] = x1 * x0; r[1] = x1 * y0; r[2] = x1;
        r[3] = y1 * x0; r[4] = y1 * y0; r[5] = y1;
        r[6] = x0;      r[7] = y0;      r[8] = 1;
    }

    /* SVD of the 7x9 system: rows 7 and 8 of the completed 9x9 basis span its null space */
    vo_or_jacobi_svd(a, 9, 7, 9, w, vt);
    double* f1 = a + 7 * 9;
    double* f2 = a + 8 * 9;

    /* F(l) = l*f1 + (1-l)*f2 = l*(f1-f2) + f2 ; det F(l) = c0 l^3 + c1 l^2 + c2 l + c3 */
    for (int i = 0; i < 9; ++i) f1[i] -= f2[i];
    double c[4], r[3] = {0, 0, 0};
    double t0, t1, t2;
    t0 = f2[4] * f2[8] - f2[5] * f2[7];
    t1 = f2[3] * f2[8] - f2[5] * f2[6];
    t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) -
           f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) -
           f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);

    int n = vo_or_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;

    for (int k = 0; k < n; ++k) {
        double* F = Fout + 9 * k;
        double lambda = r[k], mu = 1.;
        double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; F[8] = 1.; }
        else F[8] = 0.;
        for (int i = 0; i < 8; ++i) F[i] = f1[i] * lambda + f2[i] * mu;

        if (hartley) {
            /* F <- T2^T F T1 with T = [s 0 -s*cx; 0 s -s*cy; 0 0 1], then F33 -> 1 */
            double G[9], H[9];
            double A1[9] = {T1[0], 0, -T1[0] * T1[1], 0, T1[0], -T1[0] * T1[2], 0, 0, 1};
            double A2[9] = {T2[0], 0, -T2[0] * T2[1], 0, T2[0], -T2[0] * T2[2], 0, 0, 1};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    G[3 * i + j] = F[3 * i] * A1[j] + F[3 * i + 1] * A1[3 + j] + F[3 * i + 2] * A1[6 + j];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    H[3 * i + j] = A2[i] * G[j] + A2[3 + i] * G[3 + j] + A2[6 + i] * G[6 + j];
            if (fabs(H[8]) > FLT_EPSILON) {
                double inv = 1. / H[8];
                for (int i = 0; i < 9; ++i) H[i] *= inv;
            }
            memcpy(F, H, sizeof(H));
        }
    }
    return n;
}

int vo_or_find_fundamental_ransac(const float* m1, const float* m2, int n, double thr,
                                  double conf, int max_iters, int hartley,
                                  double* F9, uint8_t* mask, int32_t* info)
{
    const int mp = 7;
    info[0] = info[1] = -1; info[2] = 0; info[3] = 0;
    if (n < 15) return 0;             /* N<7: none; N in 7..14: not the RANSAC path */
    if (thr <= 0) thr = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;

    float* err = (float*)malloc(sizeof(float) * n);
    uint8_t* cur = (uint8_t*)malloc(n);
    /* the RNG stream never depends on scores: draw subsets lazily, one iteration at a time */
    int niters = max_iters > 1 ? max_iters : 1;
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * mp * (size_t)niters);
    int n_gen = vo_or_gen_subsets(n, mp, niters, 1, m1, m2, subsets);
    int max_good = 0, result = 0;
    if (n_gen == 0) goto done;

    for (int iter = 0; iter < niters; ++iter) {
        if (iter >= n_gen) break;
        float s1[14], s2[14];
        double F27[27];
        for (int i = 0; i < mp; ++i) {
            int v = subsets[iter * mp + i];
            s1[2 * i] = m1[2 * v]; s1[2 * i + 1] = m1[2 * v + 1];
            s2[2 * i] = m2[2 * v]; s2[2 * i + 1] = m2[2 * v + 1];
        }
        int nm = vo_or_run7point(s1, s2, F27, hartley);
        if (nm <= 0) continue;
        for (int k = 0; k < nm; ++k) {
            vo_or_fund_error(F27 + 9 * k, m1, m2, n, err);
            int good = vo_or_find_inliers(err, n, thr, cur);
            int fl = max_good > mp - 1 ? max_good : mp - 1;
            if (good > fl) {
                memcpy(mask, cur, n);
                memcpy(F9, F27 + 9 * k, 9 * sizeof(double));
                max_good = good;
                info[0] = iter; info[1] = k;
                niters = vo_or_update_num_iters(conf, (double)(n - good) / n, mp, niters);
            }
        }
    }
    result = max_good > 0;
done:
    info[2] = max_good; info[3] = niters;
    free(err); free(cur); free(subsets);
    return result;
}


/* findFundamentalMat with 8..14 points: OpenCV 4.6.0 takes the LMedS registrator even when the
 * caller asked for RANSAC (createLMeDSPointSetRegistrator in ptsetreg.cpp, restated
 * from its published behaviour -- OpenCV absent => parity unpinned): the iteration count is fixed up
 * front from an assumed outlier ratio of 0.45 (at least 3), every model's errors are ranked as the
 * int reinterpretation of the f32 values and the element at rank count/2 is its median, the first
 * strictly smallest median wins, inliers are the points within sigma = 2.5 * 1.4826 *
 * (1 + 5/(count - 7)) * sqrt(median) (at least 0.001).  info: best iteration, best model, inlier
 * count, iterations run.  Returns 1 when the inlier count reaches 7. */
static int cmp_i32(const void* a, const void* b)
{
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : x > y;
}

int vo_or_find_fundamental_lmeds(const float* m1, const float* m2, int n, double conf, int max_iters,
                                 int hartley, double* F9, uint8_t* mask, int32_t* info)
{
    const int mp = 7;
    info[0] = info[1] = -1; info[2] = 0; info[3] = 0;
    if (n <= mp) return 0;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    int niters = vo_or_update_num_iters(conf, 0.45, mp, max_iters);
    if (niters < 3) niters = 3;
    info[3] = niters;
    float* err = (float*)malloc(sizeof(float) * n);
    int32_t* key = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * mp * (size_t)niters);
    int n_gen = vo_or_gen_subsets_attempts(n, mp, niters, 1, m1, m2, subsets, 1000);   /* LMedS: getSubset's default maxAttempts */
    double min_median = DBL_MAX;
    int result = 0;
    for (int iter = 0; iter < niters && iter < n_gen; ++iter) {
        float s1[14], s2[14];
        double F27[27];
        for (int i = 0; i < mp; ++i) {
            int v = subsets[iter * mp + i];
            s1[2 * i] = m1[2 * v]; s1[2 * i + 1] = m1[2 * v + 1];
            s2[2 * i] = m2[2 * v]; s2[2 * i + 1] = m2[2 * v + 1];
        }
        int nm = vo_or_run7point(s1, s2, F27, hartley);
        for (int k = 0; k < nm; ++k) {
            vo_or_fund_error(F27 + 9 * k, m1, m2, n, err);
            memcpy(key, err, sizeof(float) * n);
            qsort(key, n, sizeof(int32_t), cmp_i32);
            float mf;
            memcpy(&mf, key + n / 2, sizeof(float));
            double median = mf;
            if (median < min_median) {
                min_median = median;
                memcpy(F9, F27 + 9 * k, 9 * sizeof(double));
                info[0] = iter; info[1] = k;
            }
        }
    }
    if (min_median < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - mp)) * sqrt(min_median);
        if (sigma < 0.001) sigma = 0.001;
        vo_or_fund_error(F9, m1, m2, n, err);
        info[2] = vo_or_find_inliers(err, n, sigma, mask);
        result = info[2] >= mp;
    }
    free(err); free(key); free(subsets);
    return result;
}
