/*
 * or_epnp.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * EPnP (Lepetit, Moreno-Noguer, Fua: "EPnP: An Accurate O(n) Solution to the PnP Problem",
 * IJCV 2009) in the arrangement OpenCV 4.6.0 uses for the post-RANSAC refit of
 * cv.solvePnPRansac(flags=SOLVEPNP_P3P) (reference call site src/pnp.py:16-26; SURVEY.md
 * A.5): 4 PCA control points, barycentric coordinates, 2n x 12 system M, the 4 right
 * singular vectors of M^T M with smallest singular values, beta initialisations for
 * N = 1, 2, 3, five Gauss-Newton steps each, absolute orientation, smallest mean
 * reprojection error wins.  Linear algebra goes through the oracle's Jacobi SVD.
 * OpenCV is absent here => parity unpinned.
 */
#include "or_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* A (m x n, row-major, m >= n, n <= 12) -> Ut (n x m), W (n), Vt (n x n) */
static void svd_mn(const double* A, int m, int n, double* Ut, double* W, double* Vt)
{
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < m; ++k) Ut[i * m + k] = A[k * n + i];
    vo_or_jacobi_svd(Ut, m, n, n, W, Vt);
}

/* least squares x = V diag(1/w) U^T b  (cvSolve(..., CV_SVD)); A m x n */
static void svd_solve(const double* A, int m, int n, const double* b, double* x)
{
    double Ut[12 * 12], W[12], Vt[12 * 12];
    svd_mn(A, m, n, Ut, W, Vt);
    double thr = 0;
    for (int i = 0; i < n; ++i) thr += W[i];
    thr *= 2 * 2.220446049250313e-16;
    for (int j = 0; j < n; ++j) x[j] = 0;
    for (int i = 0; i < n; ++i) {
        if (fabs(W[i]) <= thr) continue;
        double s = 0;
        for (int k = 0; k < m; ++k) s += Ut[i * m + k] * b[k];
        s /= W[i];
        for (int j = 0; j < n; ++j) x[j] += s * Vt[i * n + j];
    }
}

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* a, const double* b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

typedef struct {
    int n;
    double fu, fv, uc, vc;
    const double* pws;   /* n x 3 */
    double* us;          /* n x 2 pixels */
    double* alphas;      /* n x 4 */
    double* pcs;         /* n x 3 */
    double cws[4][3], ccs[4][3];
} epnp_t;

static void choose_control_points(epnp_t* e)
{
    int n = e->n;
    for (int j = 0; j < 3; ++j) e->cws[0][j] = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; ++j) e->cws[0][j] /= n;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        double d[3];
        for (int j = 0; j < 3; ++j) d[j] = e->pws[3 * i + j] - e->cws[0][j];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) C[3 * r + c] += d[r] * d[c];
    }
    double Ut[9], W[3], Vt[9];
    svd_mn(C, 3, 3, Ut, W, Vt);
    for (int i = 1; i < 4; ++i) {
        double k = sqrt(W[i - 1] / n);
        for (int j = 0; j < 3; ++j) e->cws[i][j] = e->cws[0][j] + k * Ut[3 * (i - 1) + j];
    }
}

static void barycentric(epnp_t* e)
{
    double cc[9], ci[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    /* inverse via SVD: solve for each unit vector */
    for (int c = 0; c < 3; ++c) {
        double b[3] = {0, 0, 0}, x[3];
        b[c] = 1;
        svd_solve(cc, 3, 3, b, x);
        for (int r = 0; r < 3; ++r) ci[3 * r + c] = x[r];
    }
    for (int i = 0; i < e->n; ++i) {
        const double* p = e->pws + 3 * i;
        double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; ++j)
            a[1 + j] = ci[3 * j] * (p[0] - e->cws[0][0]) + ci[3 * j + 1] * (p[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (p[2] - e->cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
}

static void compute_L_6x10(const double* ut, double* L)
{
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; ++i) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; ++j) {
            for (int k = 0; k < 3; ++k) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
            if (++b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; ++i) {
        double* row = L + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

static void betas_approx(const double* L, const double* rho, int which, double* betas)
{
    static const int cols1[4] = {0, 1, 3, 6}, cols2[3] = {0, 1, 2}, cols3[5] = {0, 1, 2, 3, 4};
    const int* cols = which == 1 ? cols1 : which == 2 ? cols2 : cols3;
    int nc = which == 1 ? 4 : which == 2 ? 3 : 5;
    double A[6 * 5], b[5];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < nc; ++j) A[i * nc + j] = L[10 * i + cols[j]];
    svd_solve(A, 6, nc, rho, b);
    if (which == 1) {
        if (b[0] < 0) {
            betas[0] = sqrt(-b[0]);
            betas[1] = -b[1] / betas[0]; betas[2] = -b[2] / betas[0]; betas[3] = -b[3] / betas[0];
        } else {
            betas[0] = sqrt(b[0]);
            betas[1] = b[1] / betas[0]; betas[2] = b[2] / betas[0]; betas[3] = b[3] / betas[0];
        }
        return;
    }
    if (b[0] < 0) {
        betas[0] = sqrt(-b[0]);
        betas[1] = (b[2] < 0) ? sqrt(-b[2]) : 0.0;
    } else {
        betas[0] = sqrt(b[0]);
        betas[1] = (b[2] > 0) ? sqrt(b[2]) : 0.0;
    }
    if (b[1] < 0) betas[0] = -betas[0];
    betas[2] = which == 3 ? b[3] / betas[0] : 0.0;
    betas[3] = 0.0;
}

/* Householder QR least squares for the 6 x 4 Gauss-Newton system */
static void qr_solve_6x4(double* A, double* b, double* X)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    for (int k = 0; k < nc; ++k) {
        double eta = 0;
        for (int i = k; i < nr; ++i) { double v = fabs(A[i * nc + k]); if (eta < v) eta = v; }
        if (eta == 0) { X[0] = X[1] = X[2] = X[3] = 0; return; }
        double sum2 = 0, inv_eta = 1. / eta;
        for (int i = k; i < nr; ++i) { A[i * nc + k] *= inv_eta; sum2 += A[i * nc + k] * A[i * nc + k]; }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; ++j) {
            double sum = 0;
            for (int i = k; i < nr; ++i) sum += A[i * nc + k] * A[i * nc + j];
            double tau = sum / A1[k];
            for (int i = k; i < nr; ++i) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; ++j) {
        double tau = 0;
        for (int i = j; i < nr; ++i) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        for (int i = j; i < nr; ++i) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; --i) {
        double sum = 0;
        for (int j = i + 1; j < nc; ++j) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

static void gauss_newton(const double* L, const double* rho, double* bt)
{
    for (int it = 0; it < 5; ++it) {
        double A[24], b[6], x[4] = {0, 0, 0, 0};
        for (int i = 0; i < 6; ++i) {
            const double* r = L + 10 * i;
            double* a = A + 4 * i;
            a[0] = 2 * r[0] * bt[0] + r[1] * bt[1] + r[3] * bt[2] + r[6] * bt[3];
            a[1] = r[1] * bt[0] + 2 * r[2] * bt[1] + r[4] * bt[2] + r[7] * bt[3];
            a[2] = r[3] * bt[0] + r[4] * bt[1] + 2 * r[5] * bt[2] + r[8] * bt[3];
            a[3] = r[6] * bt[0] + r[7] * bt[1] + r[8] * bt[2] + 2 * r[9] * bt[3];
            b[i] = rho[i] - (r[0] * bt[0] * bt[0] + r[1] * bt[0] * bt[1] + r[2] * bt[1] * bt[1] +
                             r[3] * bt[0] * bt[2] + r[4] * bt[1] * bt[2] + r[5] * bt[2] * bt[2] +
                             r[6] * bt[0] * bt[3] + r[7] * bt[1] * bt[3] + r[8] * bt[2] * bt[3] +
                             r[9] * bt[3] * bt[3]);
        }
        qr_solve_6x4(A, b, x);
        for (int i = 0; i < 4; ++i) bt[i] += x[i];
    }
}

static double compute_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3])
{
    int n = e->n;
    for (int i = 0; i < 4; ++i) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0;
    for (int i = 0; i < 4; ++i) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; ++i) {
        const double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; ++j)
            e->pcs[3 * i + j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < 3 * n; ++i) e->pcs[i] = -e->pcs[i];
    }
    /* absolute orientation pcs = R * pws + t */
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double* pc = e->pcs + 3 * i;
        const double* pw = e->pws + 3 * i;
        for (int j = 0; j < 3; ++j) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    double Ut[9], W[3], Vt[9];
    svd_mn(abt, 3, 3, Ut, W, Vt);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            /* R = U V^T : R[i][j] = sum_k U[i][k] V[j][k] = sum_k Ut[k][i] Vt[k][j] */
            R[i][j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] + Ut[2 * 3 + i] * Vt[2 * 3 + j];
        }
    double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                 R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
    double sum2 = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* pw = e->pws + 3 * i;
        double Xc = dot3(R[0], pw) + t[0];
        double Yc = dot3(R[1], pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc;
        double ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

/* obj n x 3 f64, img n x 2 f64 pixels.  returns 1 on success */
int vo_or_epnp(const double* obj, const double* img, int n, const double* K4, double* R9, double* t3)
{
    if (n < 4) return 0;
    epnp_t e;
    e.n = n; e.fu = K4[0]; e.fv = K4[1]; e.uc = K4[2]; e.vc = K4[3];
    e.pws = obj;
    e.us = (double*)malloc(sizeof(double) * 2 * n);
    e.alphas = (double*)malloc(sizeof(double) * 4 * n);
    e.pcs = (double*)malloc(sizeof(double) * 3 * n);
    double ifx = 1. / e.fu, ify = 1. / e.fv;
    for (int i = 0; i < n; ++i) {
        /* undistortPoints (zero distortion, f64) followed by the solver's x*fu+uc */
        double x = (img[2 * i] - e.uc) * ifx, y = (img[2 * i + 1] - e.vc) * ify;
        e.us[2 * i] = x * e.fu + e.uc;
        e.us[2 * i + 1] = y * e.fv + e.vc;
    }
    choose_control_points(&e);
    barycentric(&e);

    double* M = (double*)malloc(sizeof(double) * 2 * n * 12);
    for (int i = 0; i < n; ++i) {
        double* M1 = M + (2 * i) * 12;
        double* M2 = M1 + 12;
        const double* as = e.alphas + 4 * i;
        double u = e.us[2 * i], v = e.us[2 * i + 1];
        for (int k = 0; k < 4; ++k) {
            M1[3 * k] = as[k] * e.fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (e.uc - u);
            M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * e.fv; M2[3 * k + 2] = as[k] * (e.vc - v);
        }
    }
    double mtm[144], ut[144], d[12], vt[144];
    memset(mtm, 0, sizeof(mtm));
    for (int r = 0; r < 2 * n; ++r)
        for (int i = 0; i < 12; ++i) {
            double mi = M[r * 12 + i];
            if (mi == 0.0) continue;
            for (int j = 0; j < 12; ++j) mtm[i * 12 + j] += mi * M[r * 12 + j];
        }
    free(M);
    svd_mn(mtm, 12, 12, ut, d, vt);

    double L[60], rho[6];
    compute_L_6x10(ut, L);
    rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);

    double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
    for (int N = 1; N <= 3; ++N) {
        betas_approx(L, rho, N, Betas[N]);
        gauss_newton(L, rho, Betas[N]);
        rep[N] = compute_R_and_t(&e, ut, Betas[N], Rs[N], ts[N]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    memcpy(R9, Rs[N], 9 * sizeof(double));
    memcpy(t3, ts[N], 3 * sizeof(double));
    free(e.us); free(e.alphas); free(e.pcs);
    for (int k = 0; k < 9; ++k) if (!isfinite(R9[k])) return 0;
    return 1;
}
