/*
 * or_pnp.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * cv.solvePnPRansac(obj, img, K, dist=0, flags=SOLVEPNP_P3P, ...) + cv.Rodrigues as the
 * reference calls them at src/pnp.py:16-32, restated from SURVEY.md Appendix A.2/A.5/A.6
 * (OpenCV 4.6.0 calib3d solvepnp.cpp / p3p.cpp / epnp.cpp / calibration.cpp semantics;
 * OpenCV absent => parity unpinned):
 *   - P3P (Gao, Hou, Tang, Chang, PAMI 2003) on the first 3 of 4 sampled points, 4th point
 *     to rank, then re-ranked by the 4-point reprojection error, best one kept;
 *   - error = squared reprojection distance, f64 projection rounded to f32, f32 arithmetic;
 *   - EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) refit on all inliers of the winner.
 */
#include "or_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Rodrigues */
void vo_or_rodrigues_v2m(const double* r_in, double* R)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    double c = om_cos(theta), s = om_sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) R[k] = (c * I[k] + c1 * rrt[k]) + s * r_x[k];
}

void vo_or_rodrigues_m2v(const double* Rin, double* rvec)
{
    /* R <- U*Vt (closest rotation), via the Jacobi SVD of R */
    double At[9], W[3], Vt[9], R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) At[i * 3 + j] = Rin[j * 3 + i];
    for (int k = 0; k < 9; ++k)
        if (!(Rin[k] > -100 && Rin[k] < 100)) { rvec[0] = rvec[1] = rvec[2] = 0; return; }
    vo_or_jacobi_svd(At, 3, 3, 3, W, Vt);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += At[k * 3 + i] * Vt[k * 3 + j];
            R[i * 3 + j] = s;
        }
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = om_acos(c);
    if (s < 1e-5) {
        if (c > 0) { x = y = z = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; x = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; y = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    rvec[0] = x; rvec[1] = y; rvec[2] = z;
}

/* ------------------------------------------------------------------ projection */
static inline void project_one(const double* R, const double* t, const double* K4,
                               double X, double Y, double Z, double* u, double* v)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    *u = x * K4[0] + K4[2];
    *v = y * K4[1] + K4[3];
}

void vo_or_project_points(const double* R, const double* t, const double* K4,
                          const float* obj, int n, double* out)
{
    for (int i = 0; i < n; ++i)
        project_one(R, t, K4, obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], &out[2 * i], &out[2 * i + 1]);
}

void vo_or_pnp_error(const double* R, const double* t, const double* K4,
                     const float* obj, const float* img, int n, float* err)
{
    for (int i = 0; i < n; ++i) {
        double u, v;
        project_one(R, t, K4, obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], &u, &v);
        float pu = (float)u, pv = (float)v;
        float dx = img[2 * i] - pu, dy = img[2 * i + 1] - pv;
        float s = 0.f;
        s += dx * dx;
        s += dy * dy;
        err[i] = s;
    }
}

/* ------------------------------------------------------------------ P3P */
static int jacobi_4x4(double* A, double* D, double* U)
{
    double B[4], Z[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1. : 0.;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    memcpy(D, B, sizeof(B));
    for (int iter = 0; iter < 50; ++iter) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; ++i) {
            double* pAij = A + 5 * i + 1;
            for (int j = i + 1; j < 4; ++j) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) &&
                    fabs(D[j]) + eps_machine == fabs(D[j]))
                    *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; ++k) {
                        double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = i + 1; k <= j - 1; ++k) {
                        double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (int k = j + 1; k < 4; ++k) {
                        double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (int k = 0; k < 4; ++k) {
                        double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
                pAij++;
            }
        }
        for (int i = 0; i < 4; ++i) B[i] += Z[i];
        memcpy(D, B, sizeof(B));
        memset(Z, 0, sizeof(Z));
    }
    return 0;
}

/* absolute orientation (Horn's quaternion method) of 3 points: M_end = R * X + T */
static int p3p_align(double M_end[3][3], const double* X, double R[3][3], double T[3])
{
    double X0 = X[0], Y0 = X[1], Z0 = X[2], X1 = X[3], Y1 = X[4], Z1 = X[5], X2 = X[6], Y2 = X[7], Z2 = X[8];
    double C_start[3], C_end[3];
    for (int i = 0; i < 3; ++i) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X0 + X1 + X2) / 3;
    C_start[1] = (Y0 + Y1 + Y2) / 3;
    C_start[2] = (Z0 + Z1 + Z2) / 3;
    double s[9];
    for (int j = 0; j < 3; ++j) {
        s[0 * 3 + j] = (X0 * M_end[0][j] + X1 * M_end[1][j] + X2 * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (Y0 * M_end[0][j] + Y1 * M_end[1][j] + Y2 * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (Z0 * M_end[0][j] + Z1 * M_end[1][j] + Z2 * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    memset(Qs, 0, sizeof(Qs));
    Qs[0 * 4 + 0] = s[0] + s[4] + s[8];
    Qs[1 * 4 + 1] = s[0] - s[4] - s[8];
    Qs[2 * 4 + 2] = s[4] - s[8] - s[0];
    Qs[3 * 4 + 3] = s[8] - s[0] - s[4];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; ++i)
        if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (int i = 0; i < 4; ++i) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3); R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3); R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2); R[2][1] = 2. * (q2_3 + q0_1); R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; ++i)
        T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return 1;
}

/* distances = |BC|,|AC|,|AB| ; cosines of BPC, APC, APB ; lengths = |PA|,|PB|,|PC| */
static int p3p_lengths(double lengths[4][3], const double* distances, const double* cosines)
{
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    int n = vo_or_solve_deg4(A, B, C, D, E, roots);
    if (n == 0) return 0;
    int nb = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;
    for (int i = 0; i < n; ++i) {
        double x = roots[i];
        if (x <= 0) continue;
        double x2 = x * x;
        double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) +
               pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) +
              r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) +
             pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / sqrt(v);
        lengths[nb][0] = x * Z; lengths[nb][1] = y * Z; lengths[nb][2] = Z;
        nb++;
    }
    return nb;
}

/* mu/mv are pixel coordinates (after the undistort round trip), X = 4 object points */
static int p3p_solve(double R[4][3][3], double t[4][3], const double* K4,
                     const double* mu, const double* mv, const double* X)
{
    double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double u[4], v[4], k[3];
    for (int i = 0; i < 3; ++i) {
        u[i] = inv_fx * mu[i] - cx_fx;
        v[i] = inv_fy * mv[i] - cy_fy;
        double norm = sqrt(u[i] * u[i] + v[i] * v[i] + 1);
        k[i] = 1. / norm; u[i] *= k[i]; v[i] *= k[i];
    }
    u[3] = inv_fx * mu[3] - cx_fx;
    v[3] = inv_fy * mv[3] - cy_fy;
    double X0 = X[0], Y0 = X[1], Z0 = X[2], X1 = X[3], Y1 = X[4], Z1 = X[5], X2 = X[6], Y2 = X[7], Z2 = X[8];
    double X3 = X[9], Y3 = X[10], Z3 = X[11];
    double distances[3], cosines[3];
    distances[0] = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2) + (Z1 - Z2) * (Z1 - Z2));
    distances[1] = sqrt((X0 - X2) * (X0 - X2) + (Y0 - Y2) * (Y0 - Y2) + (Z0 - Z2) * (Z0 - Z2));
    distances[2] = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1) + (Z0 - Z1) * (Z0 - Z1));
    cosines[0] = u[1] * u[2] + v[1] * v[2] + k[1] * k[2];
    cosines[1] = u[0] * u[2] + v[0] * v[2] + k[0] * k[2];
    cosines[2] = u[0] * u[1] + v[0] * v[1] + k[0] * k[1];
    double lengths[4][3];
    int n = p3p_lengths(lengths, distances, cosines);
    int nb = 0;
    double reproj[4];
    for (int i = 0; i < n; ++i) {
        double M[3][3];
        for (int j = 0; j < 3; ++j) {
            M[j][0] = lengths[i][j] * u[j];
            M[j][1] = lengths[i][j] * v[j];
            M[j][2] = lengths[i][j] * k[j];
        }
        if (!p3p_align(M, X, R[nb], t[nb])) continue;
        double X3p = R[nb][0][0] * X3 + R[nb][0][1] * Y3 + R[nb][0][2] * Z3 + t[nb][0];
        double Y3p = R[nb][1][0] * X3 + R[nb][1][1] * Y3 + R[nb][1][2] * Z3 + t[nb][1];
        double Z3p = R[nb][2][0] * X3 + R[nb][2][1] * Y3 + R[nb][2][2] * Z3 + t[nb][2];
        double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        reproj[nb] = (mu3p - u[3]) * (mu3p - u[3]) + (mv3p - v[3]) * (mv3p - v[3]);
        nb++;
    }
    for (int i = 1; i < nb; ++i)
        for (int j = i; j > 0 && reproj[j - 1] > reproj[j]; --j) {
            double tr = reproj[j]; reproj[j] = reproj[j - 1]; reproj[j - 1] = tr;
            double tmpR[9], tmpt[3];
            memcpy(tmpR, R[j], sizeof(tmpR)); memcpy(R[j], R[j - 1], sizeof(tmpR)); memcpy(R[j - 1], tmpR, sizeof(tmpR));
            memcpy(tmpt, t[j], sizeof(tmpt)); memcpy(t[j], t[j - 1], sizeof(tmpt)); memcpy(t[j - 1], tmpt, sizeof(tmpt));
        }
    return nb;
}

int vo_or_p3p_kernel(const float* obj4, const float* img4, const double* K4,
                     double* rvec_out, double* tvec_out, double* R_out)
{
    double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    double ifx = 1. / fx, ify = 1. / fy;
    double mu[4], mv[4], X[12];
    for (int i = 0; i < 4; ++i) {
        /* undistortPoints with zero distortion: (u-cx)*(1/fx), stored back as f32; then
         * the solver's own x*fx+cx */
        float nx = (float)(((double)img4[2 * i] - cx) * ifx);
        float ny = (float)(((double)img4[2 * i + 1] - cy) * ify);
        mu[i] = nx * fx + cx;
        mv[i] = ny * fy + cy;
        X[3 * i] = obj4[3 * i]; X[3 * i + 1] = obj4[3 * i + 1]; X[3 * i + 2] = obj4[3 * i + 2];
    }
    double R[4][3][3], t[4][3];
    int ns = p3p_solve(R, t, K4, mu, mv, X);
    if (ns == 0) return 0;

    double rvecs[4][3], errs[4];
    int order[4] = {0, 1, 2, 3};
    for (int s = 0; s < ns; ++s) {
        double Rr[9];
        vo_or_rodrigues_m2v(&R[s][0][0], rvecs[s]);
        vo_or_rodrigues_v2m(rvecs[s], Rr);
        double e = 0;
        for (int i = 0; i < 4; ++i) {
            double u, v;
            project_one(Rr, t[s], K4, X[3 * i], X[3 * i + 1], X[3 * i + 2], &u, &v);
            double du = (double)img4[2 * i] - u, dv = (double)img4[2 * i + 1] - v;
            e += du * du;
            e += dv * dv;
        }
        errs[s] = e;
    }
    for (int i = 1; i < ns; ++i)
        for (int j = i; j > 0 && errs[order[j - 1]] > errs[order[j]]; --j) {
            int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
        }
    int best = order[0];
    memcpy(rvec_out, rvecs[best], 3 * sizeof(double));
    memcpy(tvec_out, t[best], 3 * sizeof(double));
    if (R_out) vo_or_rodrigues_v2m(rvecs[best], R_out);
    return ns;
}

/* ------------------------------------------------------------------ RANSAC driver */
int vo_or_solve_pnp_ransac(const float* obj, const float* img, int n, const double* K4,
                           int max_iters, float reproj_err, double conf,
                           double* rvec3, double* tvec3, uint8_t* mask, int32_t* info)
{
    const int mp = 4;
    info[0] = info[1] = -1; info[2] = 0; info[3] = 0;
    if (n < 4) return -1;
    double thr = (double)reproj_err;
    if (n == mp) {
        double R9[9];
        int ns = vo_or_p3p_kernel(obj, img, K4, rvec3, tvec3, R9);
        if (ns <= 0) return 0;
        for (int i = 0; i < n; ++i) mask[i] = 1;
        info[2] = n;
        return 2;                               /* direct solution, no refit */
    }
    int niters = max_iters > 1 ? max_iters : 1;
    int32_t* subsets = (int32_t*)malloc(sizeof(int32_t) * mp * (size_t)niters);
    float* err = (float*)malloc(sizeof(float) * n);
    uint8_t* cur = (uint8_t*)malloc(n);
    int n_gen = vo_or_gen_subsets(n, mp, niters, 0, 0, 0, subsets);
    int max_good = 0;
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    for (int iter = 0; iter < niters && iter < n_gen; ++iter) {
        float o4[12], i4[8];
        for (int i = 0; i < mp; ++i) {
            int v = subsets[iter * mp + i];
            memcpy(o4 + 3 * i, obj + 3 * v, 3 * sizeof(float));
            memcpy(i4 + 2 * i, img + 2 * v, 2 * sizeof(float));
        }
        double r3[3], t3[3], R9[9];
        if (vo_or_p3p_kernel(o4, i4, K4, r3, t3, R9) <= 0) continue;
        vo_or_pnp_error(R9, t3, K4, obj, img, n, err);
        int good = vo_or_find_inliers(err, n, thr, cur);
        int fl = max_good > mp - 1 ? max_good : mp - 1;
        if (good > fl) {
            memcpy(mask, cur, n);
            memcpy(best_r, r3, sizeof(r3)); memcpy(best_t, t3, sizeof(t3));
            max_good = good;
            info[0] = iter; info[1] = 0;
            niters = vo_or_update_num_iters(conf, (double)(n - good) / n, mp, niters);
        }
    }
    info[2] = max_good; info[3] = niters;
    free(subsets); free(err); free(cur);
    memcpy(rvec3, best_r, sizeof(best_r)); memcpy(tvec3, best_t, sizeof(best_t));
    if (max_good <= 0) return 0;

    /* EPnP refit on the inliers (f64 copies of the f32 inputs) */
    double* oi = (double*)malloc(sizeof(double) * 3 * max_good);
    double* ii = (double*)malloc(sizeof(double) * 2 * max_good);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (mask[i]) {
            oi[3 * m] = obj[3 * i]; oi[3 * m + 1] = obj[3 * i + 1]; oi[3 * m + 2] = obj[3 * i + 2];
            ii[2 * m] = img[2 * i]; ii[2 * m + 1] = img[2 * i + 1];
            m++;
        }
    double R9[9], t3[3];
    int ok = vo_or_epnp(oi, ii, m, K4, R9, t3);
    free(oi); free(ii);
    if (ok) {
        vo_or_rodrigues_m2v(R9, rvec3);
        memcpy(tvec3, t3, sizeof(t3));
        return 1;
    }
    return 0;            /* refit failed: best P3P model in rvec3/tvec3, reported failure */
}
