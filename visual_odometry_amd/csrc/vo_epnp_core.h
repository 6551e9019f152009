// vo_epnp_core.h -- the small dense algebra of EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009), shared by
// the host refit (vo_epnp.hip) and the batched device refit (vo_epnp_dev.hip): one-sided Hestenes-Jacobi
// SVD in OpenCV's sweep order and sign conventions, Householder least squares, the L matrix of the
// control-point distance constraints, closed-form beta initialisations and their Gauss-Newton polish.
#pragma once
#include <float.h>
#include <math.h>

#define VO_HD __host__ __device__ __attribute__((always_inline))

namespace vo_epnp {

// rows of `a` (n rows of length m) are orthogonalised in place; on exit row i holds
// sigma_i * u_i; v (n x n, row i = right singular vector i); sorted descending
VO_HD inline void hestenes_svd(double* a, int m, int n, double* sigma, double* v)
{
    const double tol = DBL_EPSILON * 10;
    double nrm2[12];                               // n <= 12 everywhere in EPnP
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
            double t = sigma[i]; sigma[i] = sigma[big]; sigma[big] = t;
            for (int k = 0; k < m; ++k) { t = a[i * m + k]; a[i * m + k] = a[big * m + k]; a[big * m + k] = t; }
            for (int k = 0; k < n; ++k) { t = v[i * n + k]; v[i * n + k] = v[big * n + k]; v[big * n + k] = t; }
        }
    }
}

// The same scheme for compile-time sizes with every loop but the sweep loop unrolled: identical
// arithmetic, but on the device the arrays resolve to registers instead of scratch memory.
template <int M, int N>
VO_HD inline void hestenes_fixed(double (&a)[N * M], double (&sigma)[N], double (&v)[N * N])
{
    const double tol = DBL_EPSILON * 10;
    double nrm2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) s += a[i * M + k] * a[i * M + k];
        nrm2[i] = s;
#pragma unroll
        for (int k = 0; k < N; ++k) v[i * N + k] = k == i ? 1.0 : 0.0;
    }
    constexpr int sweeps = M > 30 ? M : 30;
#pragma unroll 1
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int i = 0; i + 1 < N; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j) {
                double dot = 0;
#pragma unroll
                for (int k = 0; k < M; ++k) dot += a[i * M + k] * a[j * M + k];
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
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    const double x = c * a[i * M + k] + s * a[j * M + k], y = c * a[j * M + k] - s * a[i * M + k];
                    a[i * M + k] = x; a[j * M + k] = y;
                    ni += x * x; nj += y * y;
                }
                nrm2[i] = ni; nrm2[j] = nj;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double x = c * v[i * N + k] + s * v[j * N + k], y = c * v[j * N + k] - s * v[i * N + k];
                    v[i * N + k] = x; v[j * N + k] = y;
                }
                rotated = true;
            }
        if (!rotated) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) s += a[i * M + k] * a[i * M + k];
        sigma[i] = sqrt(s);
    }
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) {             // selection sort, first maximum wins
        int big = i;
#pragma unroll
        for (int k = i + 1; k < N; ++k)
            if (sigma[big] < sigma[k]) big = k;
#pragma unroll
        for (int q = i + 1; q < N; ++q)
            if (big == q) {
                double t = sigma[i]; sigma[i] = sigma[q]; sigma[q] = t;
#pragma unroll
                for (int k = 0; k < M; ++k) { t = a[i * M + k]; a[i * M + k] = a[q * M + k]; a[q * M + k] = t; }
#pragma unroll
                for (int k = 0; k < N; ++k) { t = v[i * N + k]; v[i * N + k] = v[q * N + k]; v[q * N + k] = t; }
            }
    }
}

// 3 x 3 symmetric matrix -> singular values and left singular vectors (rows of u)
VO_HD inline void sym_svd_u3(const double* s, double (&sigma)[3], double (&u)[9])
{
    double a[9], v[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[i * 3 + k] = s[k * 3 + i];
    hestenes_fixed<3, 3>(a, sigma, v);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double inv = sigma[i] > DBL_MIN ? 1.0 / sigma[i] : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) u[i * 3 + k] = a[i * 3 + k] * inv;
    }
}

// symmetric n x n matrix -> singular values and LEFT singular vectors (rows of u), n <= 12
VO_HD inline void sym_svd_u(const double* s, int n, double* sigma, double* u)
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
VO_HD inline bool lstsq(const double* A_in, const double* b_in, int rows, int cols, double* x)
{
    double A[6 * 5], b[6];
#pragma unroll
    for (int i = 0; i < rows * cols; ++i) A[i] = A_in[i];
#pragma unroll
    for (int i = 0; i < rows; ++i) b[i] = b_in[i];
#pragma unroll
    for (int k = 0; k < cols; ++k) {
        double nrm = 0;
#pragma unroll
        for (int i = k; i < rows; ++i) nrm += A[i * cols + k] * A[i * cols + k];
        nrm = sqrt(nrm);
        if (!(nrm > 0)) return false;
        const double alpha = A[k * cols + k] > 0 ? -nrm : nrm;
        double w[6];
#pragma unroll
        for (int i = k; i < rows; ++i) w[i] = A[i * cols + k];
        w[k] -= alpha;
        double wn = 0;
#pragma unroll
        for (int i = k; i < rows; ++i) wn += w[i] * w[i];
        if (wn > 0) {
#pragma unroll
            for (int j = k; j < cols; ++j) {
                double d = 0;
#pragma unroll
                for (int i = k; i < rows; ++i) d += w[i] * A[i * cols + j];
                d = 2 * d / wn;
#pragma unroll
                for (int i = k; i < rows; ++i) A[i * cols + j] -= d * w[i];
            }
            double d = 0;
#pragma unroll
            for (int i = k; i < rows; ++i) d += w[i] * b[i];
            d = 2 * d / wn;
#pragma unroll
            for (int i = k; i < rows; ++i) b[i] -= d * w[i];
        }
    }
#pragma unroll
    for (int i = cols - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < cols; ++j) s -= A[i * cols + j] * x[j];
        if (A[i * cols + i] == 0) return false;
        x[i] = s / A[i * cols + i];
    }
    return true;
}

VO_HD inline bool inv3(const double* m, double* o)
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

// order of the ten beta products in a row of L: 11 12 22 13 23 33 14 24 34 44
#define VO_EPNP_TABLES                                                   \
    const int kPi[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};                  \
    const int kPj[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};                  \
    const int kEdgeA[6] = {0, 0, 0, 1, 1, 2};                            \
    const int kEdgeB[6] = {1, 2, 3, 2, 3, 3};                            \
    (void)kPi; (void)kPj; (void)kEdgeA; (void)kEdgeB



VO_HD inline void build_L(const double nullv[4][12], const double cw[4][3], double L[6][10], double rho[6])
{
    VO_EPNP_TABLES;
    double dv[4][6][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int k = 0; k < 3; ++k) dv[i][e][k] = nullv[i][3 * kEdgeA[e] + k] - nullv[i][3 * kEdgeB[e] + k];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const double* x = dv[kPi[c]][e];
            const double* y = dv[kPj[c]][e];
            const double d = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
            L[e][c] = kPi[c] == kPj[c] ? d : 2.0 * d;
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double d = cw[kEdgeA[e]][k] - cw[kEdgeB[e]][k];
            s += d * d;
        }
        rho[e] = s;
    }
}

VO_HD inline void beta_start(const double L[6][10], const double rho[6], int N, double beta[4])
{
    const int sel1[4] = {0, 1, 3, 6}, sel2[3] = {0, 1, 2}, sel3[5] = {0, 1, 2, 3, 4};
    const int* sel = N == 1 ? sel1 : N == 2 ? sel2 : sel3;
    const int nc = N == 1 ? 4 : N == 2 ? 3 : 5;
    double A[30], b[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 6; ++e)
#pragma unroll
        for (int j = 0; j < nc; ++j) A[e * nc + j] = L[e][sel[j]];
    lstsq(A, rho, 6, nc, b);
    beta[0] = beta[1] = beta[2] = beta[3] = 0;
    if (N == 1) {
        const double sg = b[0] < 0 ? -1.0 : 1.0;
        beta[0] = sqrt(sg * b[0]);
#pragma unroll
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

VO_HD inline void refine_betas(const double L[6][10], const double rho[6], double beta[4])
{
    VO_EPNP_TABLES;
#pragma unroll 1
    for (int step = 0; step < 5; ++step) {
        double J[24], r[6], dx[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            double f = 0;
#pragma unroll
            for (int c = 0; c < 10; ++c) f += L[e][c] * beta[kPi[c]] * beta[kPj[c]];
            r[e] = rho[e] - f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double g = 0;
#pragma unroll
                for (int c = 0; c < 10; ++c) {
                    if (kPi[c] == k) g += L[e][c] * beta[kPj[c]];
                    if (kPj[c] == k) g += L[e][c] * beta[kPi[c]];
                }
                J[4 * e + k] = g;
            }
        }
        if (!lstsq(J, r, 6, 4, dx)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) beta[k] += dx[k];
    }
}


// absolute orientation from the cross-covariance H = sum (pc - mc)(pw - mw)^T: R = U V^T of H's SVD,
// third row flipped if det < 0; t = mc - R mw
VO_HD inline void pose_from_h(const double* H, const double* mc, const double* mw, double* R, double* t)
{
    double a[9], sig[3], v[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[3 * i + k] = H[3 * k + i];
    hestenes_fixed<3, 3>(a, sig, v);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += (sig[k] > DBL_MIN ? a[3 * k + r] / sig[k] : 0.0) * v[3 * k + c];
            R[3 * r + c] = s;
        }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int r = 0; r < 3; ++r) t[r] = mc[r] - (R[3 * r] * mw[0] + R[3 * r + 1] * mw[1] + R[3 * r + 2] * mw[2]);
}

}  // namespace vo_epnp
