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
#include <float.h>
#include <math.h>
#include <vector>

namespace {

// rows of `a` (n rows of length m) are orthogonalised in place; on exit row i holds
// sigma_i * u_i; v (n x n, row i = right singular vector i); sorted descending
void hestenes_svd(double* a, int m, int n, double* sigma, double* v)
{
    const double tol = DBL_EPSILON * 10;
    std::vector<double> nrm2(n);
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += a[i * m + k] * a[i * m + k];
        nrm2[i] = s;
        for (int k = 0; k < n; ++k) v[i * n + k] = k == i ? 1.0 : 0.0;
    }
    const int sweeps = m > 30 ? m : 30;
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        bool rotated = false;
        for (int i = 0; i + 1 < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                double* ri = a + i * m;
                double* rj = a + j * m;
                double dot = 0;
                for (int k = 0; k < m; ++k) dot += ri[k] * rj[k];
                if (fabs(dot) <= tol * sqrt(nrm2[i] * nrm2[j])) continue;
                const double p = 2 * dot, diff = nrm2[i] - nrm2[j], hyp = hypot(p, diff);
                double c, s;
                if (diff < 0) {
                    s = sqrt((hyp - diff) * 0.5 / hyp);
                    c = p / (hyp * s * 2);
                } else {
                    c = sqrt((hyp + diff) / (hyp * 2));
                    s = p / (hyp * c * 2);
                }
                double ni = 0, nj = 0;
                for (int k = 0; k < m; ++k) {
                    const double x = c * ri[k] + s * rj[k], y = c * rj[k] - s * ri[k];
                    ri[k] = x; rj[k] = y;
                    ni += x * x; nj += y * y;
                }
                nrm2[i] = ni; nrm2[j] = nj;
                double* vi = v + i * n;
                double* vj = v + j * n;
                for (int k = 0; k < n; ++k) {
                    const double x = c * vi[k] + s * vj[k], y = c * vj[k] - s * vi[k];
                    vi[k] = x; vj[k] = y;
                }
                rotated = true;
            }
        if (!rotated) break;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += a[i * m + k] * a[i * m + k];
        sigma[i] = sqrt(s);
    }
    for (int i = 0; i + 1 < n; ++i) {             // selection sort, first maximum wins
        int big = i;
        for (int k = i + 1; k < n; ++k)
            if (sigma[big] < sigma[k]) big = k;
        if (big != i) {
            std::swap(sigma[i], sigma[big]);
            for (int k = 0; k < m; ++k) std::swap(a[i * m + k], a[big * m + k]);
            for (int k = 0; k < n; ++k) std::swap(v[i * n + k], v[big * n + k]);
        }
    }
}

// symmetric n x n matrix -> singular values and LEFT singular vectors (rows of u), n <= 12
void sym_svd_u(const double* s, int n, double* sigma, double* u)
{
    double a[144], v[144];
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) a[i * n + k] = s[k * n + i];     // row i = column i of s
    hestenes_svd(a, n, n, sigma, v);
    for (int i = 0; i < n; ++i) {
        const double inv = sigma[i] > DBL_MIN ? 1.0 / sigma[i] : 0.0;
        for (int k = 0; k < n; ++k) u[i * n + k] = a[i * n + k] * inv;
    }
}

// min ||A x - b||, A rows x cols (row-major, rows >= cols, cols <= 5), Householder QR
bool lstsq(const double* A_in, const double* b_in, int rows, int cols, double* x)
{
    double A[6 * 5], b[6];
    for (int i = 0; i < rows * cols; ++i) A[i] = A_in[i];
    for (int i = 0; i < rows; ++i) b[i] = b_in[i];
    for (int k = 0; k < cols; ++k) {
        double nrm = 0;
        for (int i = k; i < rows; ++i) nrm += A[i * cols + k] * A[i * cols + k];
        nrm = sqrt(nrm);
        if (!(nrm > 0)) return false;
        const double alpha = A[k * cols + k] > 0 ? -nrm : nrm;
        double w[6];
        for (int i = k; i < rows; ++i) w[i] = A[i * cols + k];
        w[k] -= alpha;
        double wn = 0;
        for (int i = k; i < rows; ++i) wn += w[i] * w[i];
        if (wn > 0) {
            for (int j = k; j < cols; ++j) {
                double d = 0;
                for (int i = k; i < rows; ++i) d += w[i] * A[i * cols + j];
                d = 2 * d / wn;
                for (int i = k; i < rows; ++i) A[i * cols + j] -= d * w[i];
            }
            double d = 0;
            for (int i = k; i < rows; ++i) d += w[i] * b[i];
            d = 2 * d / wn;
            for (int i = k; i < rows; ++i) b[i] -= d * w[i];
        }
    }
    for (int i = cols - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < cols; ++j) s -= A[i * cols + j] * x[j];
        if (A[i * cols + i] == 0) return false;
        x[i] = s / A[i * cols + i];
    }
    return true;
}

bool inv3(const double* m, double* o)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == 0 || !isfinite(det)) return false;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

struct Problem {
    int n;
    double fu, fv, uc, vc;
    const double* pw;            // n x 3
    const double* us;            // n x 2
    std::vector<double> alpha;   // n x 4
    double cw[4][3];
    double nullv[4][12];         // singular vectors of the 4 smallest singular values, smallest first
};

// order of the ten beta products in a row of L: 11 12 22 13 23 33 14 24 34 44
const int kPi[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
const int kPj[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
const int kEdgeA[6] = {0, 0, 0, 1, 1, 2};
const int kEdgeB[6] = {1, 2, 3, 2, 3, 3};

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
    sym_svd_u(cov, 3, sig, u);
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

void build_L(const Problem& P, double L[6][10], double rho[6])
{
    double dv[4][6][3];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 6; ++e)
            for (int k = 0; k < 3; ++k) dv[i][e][k] = P.nullv[i][3 * kEdgeA[e] + k] - P.nullv[i][3 * kEdgeB[e] + k];
    for (int e = 0; e < 6; ++e) {
        for (int c = 0; c < 10; ++c) {
            const double* x = dv[kPi[c]][e];
            const double* y = dv[kPj[c]][e];
            const double d = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
            L[e][c] = kPi[c] == kPj[c] ? d : 2.0 * d;
        }
        double s = 0;
        for (int k = 0; k < 3; ++k) {
            const double d = P.cw[kEdgeA[e]][k] - P.cw[kEdgeB[e]][k];
            s += d * d;
        }
        rho[e] = s;
    }
}

void beta_start(const double L[6][10], const double rho[6], int N, double beta[4])
{
    static const int sel1[4] = {0, 1, 3, 6}, sel2[3] = {0, 1, 2}, sel3[5] = {0, 1, 2, 3, 4};
    const int* sel = N == 1 ? sel1 : N == 2 ? sel2 : sel3;
    const int nc = N == 1 ? 4 : N == 2 ? 3 : 5;
    double A[30], b[5] = {0, 0, 0, 0, 0};
    for (int e = 0; e < 6; ++e)
        for (int j = 0; j < nc; ++j) A[e * nc + j] = L[e][sel[j]];
    lstsq(A, rho, 6, nc, b);
    beta[0] = beta[1] = beta[2] = beta[3] = 0;
    if (N == 1) {
        const double sg = b[0] < 0 ? -1.0 : 1.0;
        beta[0] = sqrt(sg * b[0]);
        for (int k = 1; k < 4; ++k) beta[k] = sg * b[k] / beta[0];
        return;
    }
    if (b[0] < 0) {
        beta[0] = sqrt(-b[0]);
        beta[1] = b[2] < 0 ? sqrt(-b[2]) : 0.0;
    } else {
        beta[0] = sqrt(b[0]);
        beta[1] = b[2] > 0 ? sqrt(b[2]) : 0.0;
    }
    if (b[1] < 0) beta[0] = -beta[0];
    if (N == 3) beta[2] = b[3] / beta[0];
}

void refine_betas(const double L[6][10], const double rho[6], double beta[4])
{
    for (int step = 0; step < 5; ++step) {
        double J[24], r[6], dx[4] = {0, 0, 0, 0};
        for (int e = 0; e < 6; ++e) {
            double f = 0;
            for (int c = 0; c < 10; ++c) f += L[e][c] * beta[kPi[c]] * beta[kPj[c]];
            r[e] = rho[e] - f;
            for (int k = 0; k < 4; ++k) {
                double g = 0;
                for (int c = 0; c < 10; ++c) {
                    if (kPi[c] == k) g += L[e][c] * beta[kPj[c]];
                    if (kPj[c] == k) g += L[e][c] * beta[kPi[c]];
                }
                J[4 * e + k] = g;
            }
        }
        if (!lstsq(J, r, 6, 4, dx)) return;
        for (int k = 0; k < 4; ++k) beta[k] += dx[k];
    }
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
    // H = U S V^T  ->  R = U V^T.  rows of `a` below end up as sigma_k * u_k, rows of v as v_k
    double a[9], sig[3], v[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) a[3 * i + k] = H[3 * k + i];
    hestenes_svd(a, 3, 3, sig, v);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += (sig[k] > DBL_MIN ? a[3 * k + r] / sig[k] : 0.0) * v[3 * k + c];
            R[3 * r + c] = s;
        }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int r = 0; r < 3; ++r) t[r] = mc[r] - (R[3 * r] * mw[0] + R[3 * r + 1] * mw[1] + R[3 * r + 2] * mw[2]);

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
    build_L(P, L, rho);

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
