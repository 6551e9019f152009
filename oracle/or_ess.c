/*
 * or_ess.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * cv2.findEssentialMat(p1, p2, K, cv2.RANSAC, prob, threshold, maxIters) as wrapped by the
 * reference at src/utils/geometry.py:43-45 (dormant: no caller), and cv.triangulatePoints
 * (reference src/utils/disambiguate_relative_pose.py:42), restated from SURVEY.md Appendix
 * A.2/A.4/A.7 (OpenCV 4.6.0 five-point.cpp / triangulate.cpp semantics; OpenCV absent =>
 * parity unpinned):
 *   - Nister's five-point algorithm (PAMI 2004): 4-D null space of the 5 x 9 epipolar
 *     system, the ten cubic constraints det(E)=0 and 2EE^T E - tr(EE^T)E = 0, Gauss-Jordan
 *     on the 10 x 20 coefficient matrix, 3 x 3 polynomial matrix B(z), det B(z) = 0 of
 *     degree 10, real roots by Durand-Kerner (cv::solvePoly), E normalised to unit
 *     Frobenius norm;
 *   - Sampson error in f64 on K-normalised f64 points, rounded to f32.
 * The polynomial algebra is carried out generically here (OpenCV uses generated
 * expressions), so models agree with OpenCV's only to rounding.
 */
#include "or_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void vo_or_sampson_error(const double* E, const double* x1, const double* x2, int n, float* err)
{
    for (int i = 0; i < n; ++i) {
        double ax = x1[2 * i], ay = x1[2 * i + 1], bx = x2[2 * i], by = x2[2 * i + 1];
        /* Matx33 * Vec3: s = 0; s += a(i,k)*b(k) */
        double Ex1_0 = E[0] * ax + E[1] * ay + E[2] * 1.;
        double Ex1_1 = E[3] * ax + E[4] * ay + E[5] * 1.;
        double Ex1_2 = E[6] * ax + E[7] * ay + E[8] * 1.;
        double Etx2_0 = E[0] * bx + E[3] * by + E[6] * 1.;
        double Etx2_1 = E[1] * bx + E[4] * by + E[7] * 1.;
        double x2tEx1 = bx * Ex1_0 + by * Ex1_1 + 1. * Ex1_2;
        double a = Ex1_0 * Ex1_0, b = Ex1_1 * Ex1_1, c = Etx2_0 * Etx2_0, d = Etx2_1 * Etx2_1;
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
}

/* ---- polynomials in (x,y,z) of total degree <= 3, dense over exponents -------------- */
typedef struct { double c[4][4][4]; } poly3;     /* c[i][j][k] of x^i y^j z^k */

static void p_zero(poly3* p) { memset(p, 0, sizeof(*p)); }
static void p_mul(const poly3* a, const poly3* b, poly3* r)
{
    p_zero(r);
    for (int i = 0; i < 4; ++i) for (int j = 0; j + i < 4; ++j) for (int k = 0; k + i + j < 4; ++k) {
        double av = a->c[i][j][k];
        if (av == 0) continue;
        for (int l = 0; l + i < 4; ++l) for (int m = 0; m + j < 4; ++m) for (int n = 0; n + k < 4; ++n) {
            if (i + j + k + l + m + n > 3) continue;
            r->c[i + l][j + m][k + n] += av * b->c[l][m][n];
        }
    }
}
static void p_axpy(poly3* r, double s, const poly3* a)
{
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k)
        r->c[i][j][k] += s * a->c[i][j][k];
}

/* monomial order: [x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1] */
static const int MON[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};

static void p_row(const poly3* p, double* row)
{
    for (int m = 0; m < 20; ++m) row[m] = p->c[MON[m][0]][MON[m][1]][MON[m][2]];
}

/* univariate polynomials in z, ascending coefficients */
static void z_mul(const double* a, int da, const double* b, int db, double* r)
{
    for (int i = 0; i <= da + db; ++i) r[i] = 0;
    for (int i = 0; i <= da; ++i) for (int j = 0; j <= db; ++j) r[i + j] += a[i] * b[j];
}

int vo_or_run5point(const double* q1, const double* q2, double* Eout)
{
    /* 5 x 9 system rows [x2x1 x2y1 x2 y2x1 y2y1 y2 x1 y1 1] => x2^T E x1 = 0, E row-major */
    double A9[9 * 9], w[5], vt[25];
    memset(A9, 0, sizeof(A9));
    for (int i = 0; i < 5; ++i) {
        double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        double* r = A9 + 9 * i;
        r[0] = x2 * x1; r[1] = x2 * y1; r[2] = x2;
        r[3] = y2 * x1; r[4] = y2 * y1; r[5] = y2;
        r[6] = x1; r[7] = y1; r[8] = 1.0;
    }
    vo_or_jacobi_svd(A9, 9, 5, 9, w, vt);
    const double* Nb[4] = {A9 + 5 * 9, A9 + 6 * 9, A9 + 7 * 9, A9 + 8 * 9};   /* X Y Z W */

    /* E entries as linear polynomials */
    poly3 Ep[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        p_zero(&Ep[r][c]);
        Ep[r][c].c[1][0][0] = Nb[0][3 * r + c];
        Ep[r][c].c[0][1][0] = Nb[1][3 * r + c];
        Ep[r][c].c[0][0][1] = Nb[2][3 * r + c];
        Ep[r][c].c[0][0][0] = Nb[3][3 * r + c];
    }
    /* EEt and trace */
    poly3 EEt[3][3], tr, tmp, tmp2;
    p_zero(&tr);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        p_zero(&EEt[r][c]);
        for (int k = 0; k < 3; ++k) { p_mul(&Ep[r][k], &Ep[c][k], &tmp); p_axpy(&EEt[r][c], 1.0, &tmp); }
    }
    for (int r = 0; r < 3; ++r) p_axpy(&tr, 1.0, &EEt[r][r]);

    double A[10][20];
    /* rows 0..8: 2 EEt E - tr E */
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        poly3 acc; p_zero(&acc);
        for (int k = 0; k < 3; ++k) { p_mul(&EEt[r][k], &Ep[k][c], &tmp); p_axpy(&acc, 2.0, &tmp); }
        p_mul(&tr, &Ep[r][c], &tmp);
        p_axpy(&acc, -1.0, &tmp);
        p_row(&acc, A[3 * r + c]);
    }
    /* row 9: det E */
    {
        poly3 acc; p_zero(&acc);
        p_mul(&Ep[1][1], &Ep[2][2], &tmp); p_mul(&Ep[0][0], &tmp, &tmp2); p_axpy(&acc, 1.0, &tmp2);
        p_mul(&Ep[1][2], &Ep[2][1], &tmp); p_mul(&Ep[0][0], &tmp, &tmp2); p_axpy(&acc, -1.0, &tmp2);
        p_mul(&Ep[1][0], &Ep[2][2], &tmp); p_mul(&Ep[0][1], &tmp, &tmp2); p_axpy(&acc, -1.0, &tmp2);
        p_mul(&Ep[1][2], &Ep[2][0], &tmp); p_mul(&Ep[0][1], &tmp, &tmp2); p_axpy(&acc, 1.0, &tmp2);
        p_mul(&Ep[1][0], &Ep[2][1], &tmp); p_mul(&Ep[0][2], &tmp, &tmp2); p_axpy(&acc, 1.0, &tmp2);
        p_mul(&Ep[1][1], &Ep[2][0], &tmp); p_mul(&Ep[0][2], &tmp, &tmp2); p_axpy(&acc, -1.0, &tmp2);
        p_row(&acc, A[9]);
    }
    /* Gauss-Jordan with partial pivoting on the first 10 columns */
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        double best = fabs(A[col][col]);
        for (int r = col + 1; r < 10; ++r) if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); piv = r; }
        if (best < DBL_MIN) return 0;
        if (piv != col) for (int k = 0; k < 20; ++k) { double t = A[col][k]; A[col][k] = A[piv][k]; A[piv][k] = t; }
        double inv = 1.0 / A[col][col];
        for (int k = 0; k < 20; ++k) A[col][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            double f = A[r][col];
            if (f == 0) continue;
            for (int k = 0; k < 20; ++k) A[r][k] -= f * A[col][k];
        }
    }
    /* B(z): rows from (x2z,x2), (y2z,y2), (xyz,xy): row(a) - z*row(b); columns
     * [x: deg3][y: deg3][1: deg4], ascending coefficient arrays */
    double B[3][3][5];
    memset(B, 0, sizeof(B));
    for (int i = 0; i < 3; ++i) {
        const double* ra = A[4 + 2 * i] + 10;     /* tail: xz2 xz x yz2 yz y z3 z2 z 1 */
        const double* rb = A[5 + 2 * i] + 10;
        /* x column: ra: z^2,z^1,z^0 = ra[0],ra[1],ra[2] */
        B[i][0][0] = ra[2];          B[i][0][1] = ra[1] - rb[2]; B[i][0][2] = ra[0] - rb[1]; B[i][0][3] = -rb[0];
        B[i][1][0] = ra[5];          B[i][1][1] = ra[4] - rb[5]; B[i][1][2] = ra[3] - rb[4]; B[i][1][3] = -rb[3];
        B[i][2][0] = ra[9];          B[i][2][1] = ra[8] - rb[9]; B[i][2][2] = ra[7] - rb[8];
        B[i][2][3] = ra[6] - rb[7];  B[i][2][4] = -rb[6];
    }
    /* det B(z), degree 10 */
    double c[11], t1[8], t2[8], t3[12];
    memset(c, 0, sizeof(c));
    #define MINOR(r1, c1, d1, r2, c2, d2, out) z_mul(B[r1][c1], d1, B[r2][c2], d2, out)
    {
        /* + B00 (B11 B22 - B12 B21) - B01 (B10 B22 - B12 B20) + B02 (B10 B21 - B11 B20) */
        double m[8];
        MINOR(1, 1, 3, 2, 2, 4, t1); MINOR(1, 2, 4, 2, 1, 3, t2);
        for (int i = 0; i < 8; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][0], 3, m, 7, t3);
        for (int i = 0; i < 11; ++i) c[i] += t3[i];
        MINOR(1, 0, 3, 2, 2, 4, t1); MINOR(1, 2, 4, 2, 0, 3, t2);
        for (int i = 0; i < 8; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][1], 3, m, 7, t3);
        for (int i = 0; i < 11; ++i) c[i] -= t3[i];
        MINOR(1, 0, 3, 2, 1, 3, t1); MINOR(1, 1, 3, 2, 0, 3, t2);
        t1[7] = t2[7] = 0;
        for (int i = 0; i < 7; ++i) m[i] = t1[i] - t2[i];
        z_mul(B[0][2], 4, m, 6, t3);
        for (int i = 0; i < 11; ++i) c[i] += t3[i];
    }
    #undef MINOR
    double re[10], im[10];
    int nr = vo_or_solve_poly(c, 10, re, im);
    int count = 0;
    for (int i = 0; i < nr && count < 10; ++i) {
        if (fabs(im[i]) > 1e-10) continue;
        double z1 = re[i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; ++j) {
            bz[3 * j] = B[j][0][3] * z3 + B[j][0][2] * z2 + B[j][0][1] * z1 + B[j][0][0];
            bz[3 * j + 1] = B[j][1][3] * z3 + B[j][1][2] * z2 + B[j][1][1] * z1 + B[j][1][0];
            bz[3 * j + 2] = B[j][2][4] * z4 + B[j][2][3] * z3 + B[j][2][2] * z2 + B[j][2][1] * z1 + B[j][2][0];
        }
        /* null vector of Bz: right singular vector of the smallest singular value */
        double At[9], W3[3], Vt3[9];
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) At[cc * 3 + r] = bz[r * 3 + cc];
        vo_or_jacobi_svd(At, 3, 3, 3, W3, Vt3);
        const double* xy1 = Vt3 + 6;
        if (fabs(xy1[2]) < 1e-10) continue;
        double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; ++k) {
            Ev[k] = Nb[0][k] * x + Nb[1][k] * y + Nb[2][k] * z1 + Nb[3][k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; ++k) Eout[9 * count + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

int vo_or_find_essential_ransac(const double* p1, const double* p2, int n, const double* K9,
                                double prob, double thr, int max_iters,
                                double* E9, uint8_t* mask, int32_t* info)
{
    const int mp = 5;
    info[0] = info[1] = -1; info[2] = 0; info[3] = 0;
    if (n < mp) return 0;
    double fx = K9[0], fy = K9[4], cx = K9[2], cy = K9[5];
    double ifx = 1. / fx, ify = 1. / fy;
    double* x1 = (double*)malloc(sizeof(double) * 2 * n);
    double* x2 = (double*)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < n; ++i) {
        x1[2 * i] = p1[2 * i] * ifx + (-cx * ifx);  x1[2 * i + 1] = p1[2 * i + 1] * ify + (-cy * ify);
        x2[2 * i] = p2[2 * i] * ifx + (-cx * ifx);  x2[2 * i + 1] = p2[2 * i + 1] * ify + (-cy * ify);
    }
    thr /= (fx + fy) / 2;
    int result = 0;
    if (n == mp) {
        double E90[90];
        int nm = vo_or_run5point(x1, x2, E90);
        if (nm > 0) { memcpy(E9, E90, 9 * sizeof(double)); for (int i = 0; i < n; ++i) mask[i] = 1; result = nm; }
        free(x1); free(x2);
        return result;
    }
    int niters = max_iters > 1 ? max_iters : 1;
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * mp * (size_t)niters);
    float* err = (float*)malloc(sizeof(float) * n);
    uint8_t* cur = (uint8_t*)malloc(n);
    int n_gen = vo_or_gen_subsets(n, mp, niters, 0, 0, 0, subsets);
    int max_good = 0;
    for (int iter = 0; iter < niters && iter < n_gen; ++iter) {
        double s1[10], s2[10], E90[90];
        for (int i = 0; i < mp; ++i) {
            int v = subsets[iter * mp + i];
            s1[2 * i] = x1[2 * v]; s1[2 * i + 1] = x1[2 * v + 1];
            s2[2 * i] = x2[2 * v]; s2[2 * i + 1] = x2[2 * v + 1];
        }
        int nm = vo_or_run5point(s1, s2, E90);
        for (int k = 0; k < nm; ++k) {
            vo_or_sampson_error(E90 + 9 * k, x1, x2, n, err);
            int good = vo_or_find_inliers(err, n, thr, cur);
            int fl = max_good > mp - 1 ? max_good : mp - 1;
            if (good > fl) {
                memcpy(mask, cur, n);
                memcpy(E9, E90 + 9 * k, 9 * sizeof(double));
                max_good = good;
                info[0] = iter; info[1] = k;
                niters = vo_or_update_num_iters(prob, (double)(n - good) / n, mp, niters);
            }
        }
    }
    info[2] = max_good; info[3] = niters;
    free(subsets); free(err); free(cur); free(x1); free(x2);
    return max_good > 0;
}

/* ------------------------------------------------------------------ triangulation */
/* P1,P2: 3x4 row-major; x1,x2: n x 2; X4: n x 4 homogeneous (not normalised) */
void vo_or_triangulate(const double* P1, const double* P2, const double* x1,
                       const double* x2, int n, double* X4)
{
    for (int i = 0; i < n; ++i) {
        double A[16];
        const double* P[2] = {P1, P2};
        const double* xs[2] = {x1 + 2 * i, x2 + 2 * i};
        for (int j = 0; j < 2; ++j) {
            double x = xs[j][0], y = xs[j][1];
            for (int k = 0; k < 4; ++k) {
                A[(2 * j) * 4 + k] = x * P[j][8 + k] - P[j][k];
                A[(2 * j + 1) * 4 + k] = y * P[j][8 + k] - P[j][4 + k];
            }
        }
        double At[16], W[4], Vt[16];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) At[c * 4 + r] = A[r * 4 + c];
        vo_or_jacobi_svd(At, 4, 4, 4, W, Vt);
        for (int k = 0; k < 4; ++k) X4[4 * i + k] = Vt[12 + k];
    }
}
