// vo_epnp.hip -- host-only: the EPnP refit cv.solvePnPRansac(flags=SOLVEPNP_P3P) runs on the
// inliers of the winning P3P model (reference src/pnp.py:16-26).  It is
// serial, O(inliers) and runs once per call, so it stays on the CPU next to the GPU RANSAC batch.
//
// Method: Lepetit, Moreno-Noguer, Fua, "EPnP: An Accurate O(n) Solution to the PnP Problem"
// (IJCV 2009): control points from the PCA of the object points, barycentric coordinates, the
// 12x12 normal matrix of the projection constraints, its four smallest singular vectors, closed
// form beta initialisations for N = 1, 2, 3, five Gauss-Newton steps each, absolute orientation,
// lowest mean reprojection error wins.
//
// The result of EPnP under noise depends on the SIGNS of the PCA axes (they place the control
// points), so the SVD here is the one-sided Hestenes-Jacobi scheme OpenCV's own SVD uses (V
// starts as the identity, Givens pair (i,j) swept in row order, singular triplets sorted by
// descending value): same algorithm, same signs.  Least-squares sub-problems have unique
// solutions and go through a plain Householder QR.
#include "vo_common.h"
#include "vo_epnp_core.h"
#include <vector>

namespace {

using namespace vo_epnp;

struct Problem {
    int n;
    double fu, fv, uc, vc;
    const double* pw;            // n x 3
    const double* us;            // n x 2
    std::vector<double> alpha;   // n x 4
    double cw[4][3];
    double nullv[4][12];         // singular vectors of the 4 smallest singular values, smallest first
};

void control_points(Problem& P)
{
    const int n = P.n;
    double c0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) c0[j] += P.pw[3 * i + j];
    for (int j = 0; j < 3; ++j) c0[j] /= n;
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double d[3] = {P.pw[3 * i] - c0[0], P.pw[3 * i + 1] - c0[1], P.pw[3 * i + 2] - c0[2]};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[3 * r + c] += d[r] * d[c];
    }
    double sig[3], u[9];
    sym_svd_u3(cov, sig, u);
    for (int j = 0; j < 3; ++j) P.cw[0][j] = c0[j];
    for (int a = 0; a < 3; ++a) {
        const double k = sqrt(sig[a] / n);
        for (int j = 0; j < 3; ++j) P.cw[a + 1][j] = c0[j] + k * u[3 * a + j];
    }
}

bool barycentric(Problem& P)
{
    double basis[9], inv[9];
    for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) basis[3 * r + a] = P.cw[a + 1][r] - P.cw[0][r];
    if (!inv3(basis, inv)) return false;
    P.alpha.resize((size_t)4 * P.n);
    for (int i = 0; i < P.n; ++i) {
        const double d[3] = {P.pw[3 * i] - P.cw[0][0], P.pw[3 * i + 1] - P.cw[0][1], P.pw[3 * i + 2] - P.cw[0][2]};
        double* a = &P.alpha[4 * (size_t)i];
        for (int j = 0; j < 3; ++j) a[1 + j] = inv[3 * j] * d[0] + inv[3 * j + 1] * d[1] + inv[3 * j + 2] * d[2];
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    return true;
}

void null_space(Problem& P)
{
    double mtm[144];
    for (int i = 0; i < 144; ++i) mtm[i] = 0;
    for (int i = 0; i < P.n; ++i) {
        const double* a = &P.alpha[4 * (size_t)i];
        const double du = P.uc - P.us[2 * i], dv = P.vc - P.us[2 * i + 1];
        double r1[12], r2[12];
        for (int k = 0; k < 4; ++k) {
            r1[3 * k] = a[k] * P.fu; r1[3 * k + 1] = 0; r1[3 * k + 2] = a[k] * du;
            r2[3 * k] = 0; r2[3 * k + 1] = a[k] * P.fv; r2[3 * k + 2] = a[k] * dv;
        }
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 12; ++c) mtm[12 * r + c] += r1[r] * r1[c];
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 12; ++c) mtm[12 * r + c] += r2[r] * r2[c];
    }
    double sig[12], u[144];
    sym_svd_u(mtm, 12, sig, u);
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 12; ++k) P.nullv[i][k] = u[12 * (11 - i) + k];
}

double pose_from_betas(const Problem& P, const double beta[4], double R[9], double t[3], std::vector<double>& pc)
{
    const int n = P.n;
    double cc[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 3; ++k) cc[j][k] += beta[i] * P.nullv[i][3 * j + k];
    pc.resize((size_t)3 * n);
    for (int i = 0; i < n; ++i) {
        const double* a = &P.alpha[4 * (size_t)i];
        for (int k = 0; k < 3; ++k) pc[3 * (size_t)i + k] = a[0] * cc[0][k] + a[1] * cc[1][k] + a[2] * cc[2][k] + a[3] * cc[3][k];
    }
    if (pc[2] < 0)
        for (size_t i = 0; i < pc.size(); ++i) pc[i] = -pc[i];

    double mc[3] = {0, 0, 0}, mw[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { mc[k] += pc[3 * (size_t)i + k]; mw[k] += P.pw[3 * i + k]; }
    for (int k = 0; k < 3; ++k) { mc[k] /= n; mw[k] /= n; }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // sum (pc - mc)(pw - mw)^T
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[3 * r + c] += (pc[3 * (size_t)i + r] - mc[r]) * (P.pw[3 * i + c] - mw[c]);
    pose_from_h(H, mc, mw, R, t);

    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double* w = P.pw + 3 * i;
        const double X = R[0] * w[0] + R[1] * w[1] + R[2] * w[2] + t[0];
        const double Y = R[3] * w[0] + R[4] * w[1] + R[5] * w[2] + t[1];
        const double iz = 1.0 / (R[6] * w[0] + R[7] * w[1] + R[8] * w[2] + t[2]);
        const double du = P.us[2 * i] - (P.uc + P.fu * X * iz), dv = P.us[2 * i + 1] - (P.vc + P.fv * Y * iz);
        sum += sqrt(du * du + dv * dv);
    }
    return sum / n;
}

}  // namespace

extern "C" int vo_host_epnp(const double* obj_host, const double* img_host, int n, const double* K4_host,
                            double* R9_host, double* t3_host)
{
    VO_CHECK_ARG(obj_host && img_host && K4_host && R9_host && t3_host, VO_EINVAL, "vo_host_epnp: null pointer");
    VO_CHECK_ARG(n >= 4, VO_EINVAL, "vo_host_epnp: needs at least 4 correspondences, got %d", n);
    Problem P;
    P.n = n;
    P.fu = K4_host[0]; P.fv = K4_host[1]; P.uc = K4_host[2]; P.vc = K4_host[3];
    P.pw = obj_host;
    P.us = img_host;
    control_points(P);
    if (!barycentric(P)) return 0;
    null_space(P);
    double L[6][10], rho[6];
    build_L(P.nullv, P.cw, L, rho);

    double best = 0;
    bool have = false;
    std::vector<double> pc;
    for (int N = 1; N <= 3; ++N) {
        double beta[4], R[9], t[3];
        beta_start(L, rho, N, beta);
        refine_betas(L, rho, beta);
        const double err = pose_from_betas(P, beta, R, t, pc);
        if (!have || err < best) {
            have = true;
            best = err;
            for (int k = 0; k < 9; ++k) R9_host[k] = R[k];
            for (int k = 0; k < 3; ++k) t3_host[k] = t[k];
        }
    }
    for (int k = 0; k < 9; ++k)
        if (!isfinite(R9_host[k])) return 0;
    for (int k = 0; k < 3; ++k)
        if (!isfinite(t3_host[k])) return 0;
    return 1;
}
