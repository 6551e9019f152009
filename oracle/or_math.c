/*
 * or_math.c -- CPU ORACLE (test infrastructure, NOT product code; see vo_oracle.h).
 *
 * Numerical building blocks that OpenCV 4.6.0's calib3d routines rest on, restated from
 * the published algorithms (OpenCV absent here => parity unpinned):
 *   - one-sided (Hestenes) Jacobi SVD incl. the "complete the basis from a fixed
 *     pseudo-random +-1/m vector by Gram-Schmidt" step used for FULL_UV   [core/lapack.cpp]
 *   - cv::solveCubic (Cardano / trigonometric form)                       [core/mathfuncs.cpp]
 *   - cv::solvePoly (Durand-Kerner)                                       [core/mathfuncs.cpp]
 *   - solve_deg2/3/4 of calib3d's polynom_solver (MathWorld formulas)     [calib3d/polynom_solver.cpp]
 *
 * Elementary functions go through om_* so that "det-math" mode (vo_or_set_det_math(1))
 * swaps libm for routines made of IEEE + - * / sqrt only; the HIP device code uses the
 * same routines in the same operation order (visual_odometry_amd/csrc/vo_detmath.h is
 * written independently from the same specification: fdlibm's published kernels).
 */
#include "or_internal.h"
#include <float.h>
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ det-math */
/* cos/sin kernels on |r| <= pi/4 and the acos rational: coefficients are fdlibm's
 * (Sun Microsystems, freely distributable), evaluated in plain Horner form. */
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
static const double PIO2_HI = 1.57079632673412561417e+00; /* 33 bits of pi/2 */
static const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */

static double k_sin(double x)
{
    double z = x * x;
    double p = S6;
    p = p * z + S5; p = p * z + S4; p = p * z + S3; p = p * z + S2; p = p * z + S1;
    return x + x * (z * p);
}
static double k_cos(double x)
{
    double z = x * x;
    double p = C6;
    p = p * z + C5; p = p * z + C4; p = p * z + C3; p = p * z + C2; p = p * z + C1;
    return (1.0 - 0.5 * z) + z * (z * p);
}
/* valid for 0 <= x < ~100 (k small enough that k*PIO2_HI is exact) */
static void det_sincos(double x, double* s, double* c)
{
    double kf = floor(x * 6.36619772367581382433e-01 + 0.5);
    int k = (int)kf;
    double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    double sr = k_sin(r), cr = k_cos(r);
    switch (k & 3) {
        case 0: *s = sr;  *c = cr;  break;
        case 1: *s = cr;  *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
static double det_acos(double x)
{
    static const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                        pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                        pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                        qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    static const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                        pi = 3.14159265358979311600e+00;
    double ax = fabs(x);
    if (ax >= 1.0) return x > 0 ? 0.0 : pi;          /* callers clamp to [-1,1] */
    double z = ax < 0.5 ? x * x : (1.0 - ax) * 0.5;
    double p = pS5;
    p = p * z + pS4; p = p * z + pS3; p = p * z + pS2; p = p * z + pS1; p = p * z + pS0;
    p = p * z;
    double q = qS4;
    q = q * z + qS3; q = q * z + qS2; q = q * z + qS1; q = q * z + 1.0;
    double r = p / q;
    if (ax < 0.5) return pio2_hi - (x - (pio2_lo - x * r));
    double s = sqrt(z);
    double w = s + s * r;
    return x > 0 ? 2.0 * w : pi - 2.0 * w;
}
/* cube root of x >= 0: exponent split + 6 Newton steps (fixed count => deterministic) */
static double det_cbrt(double x)
{
    if (!(x > 0.0)) return 0.0;
    int e;
    double m = frexp(x, &e);              /* m in [0.5,1) */
    int r = e % 3;
    if (r < 0) r += 3;
    if (r) { m = ldexp(m, r - 3); e += 3 - r; }   /* m in [0.125,1), e % 3 == 0 */
    double y = 0.4285714285714286 + 0.5714285714285714 * m;
    for (int i = 0; i < 6; ++i) {
        double y2 = y * y;
        y = y - (y2 * y - m) / (3.0 * y2);
    }
    return ldexp(y, e / 3);
}

double om_acos(double x) { return g_or_det() ? det_acos(x) : acos(x); }
double om_cos(double x)
{
    if (!g_or_det()) return cos(x);
    double s, c; det_sincos(x, &s, &c); return c;
}
double om_sin(double x)
{
    if (!g_or_det()) return sin(x);
    double s, c; det_sincos(x, &s, &c); return s;
}
double om_cbrt_pow(double x) { return g_or_det() ? det_cbrt(x) : pow(x, 1. / 3); }
double om_hypot(double a, double b) { return g_or_det() ? sqrt(a * a + b * b) : hypot(a, b); }

/* ------------------------------------------------------------------ Jacobi SVD */
/* At: n1 rows of length m (rows n..n1-1 must be zero on entry when n1 > n).  On exit the
 * first n rows are the left singular vectors of the m x n matrix whose columns were the
 * input rows (sorted by singular value, descending), rows n..n1-1 complete the basis.
 * W[n] singular values, Vt n x n. */
void vo_or_jacobi_svd(double* At, int m, int n, int n1, double* W_out, double* Vt)
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    double W[16];
    int max_iter = m > 30 ? m : 30;

    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sd;
        if (Vt) {
            for (int k = 0; k < n; ++k) Vt[i * n + k] = 0;
            Vt[i * n + i] = 1;
        }
    }

    for (int iter = 0; iter < max_iter; ++iter) {
        int changed = 0;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;

                p *= 2;
                double beta = a - b, gamma = om_hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; ++k) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * n, *Vj = Vt + j * n;
                    for (int k = 0; k < n; ++k) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }

    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; ++i) {
        int j = i;
        for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (Vt) {
                for (int k = 0; k < m; ++k) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
                for (int k = 0; k < n; ++k) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
            }
        }
    }
    for (int i = 0; i < n; ++i) W_out[i] = W[i];
    if (!Vt) return;

    uint64_t rs = 0x12345678ULL;     /* cv::RNG(0x12345678) */
    for (int i = 0; i < n1; ++i) {
        double sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
            const double val0 = 1. / m;
            for (int k = 0; k < m; ++k) {
                rs = (uint64_t)(uint32_t)rs * 4164903690u + (uint32_t)(rs >> 32);
                At[i * m + k] = ((uint32_t)rs & 256) != 0 ? val0 : -val0;
            }
            for (int it = 0; it < 2; ++it)
                for (int j = 0; j < i; ++j) {
                    sd = 0;
                    for (int k = 0; k < m; ++k) sd += At[i * m + k] * At[j * m + k];
                    double asum = 0;
                    for (int k = 0; k < m; ++k) {
                        double t = At[i * m + k] - sd * At[j * m + k];
                        At[i * m + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; ++k) At[i * m + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < m; ++k) { double t = At[i * m + k]; sd += t * t; }
            sd = sqrt(sd);
        }
        double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; ++k) At[i * m + k] *= s;
    }
}

/* ------------------------------------------------------------------ cv::solveCubic */
/* c4 = {a0,a1,a2,a3}: a0 x^3 + a1 x^2 + a2 x + a3 = 0.  returns #real roots (-1: all x) */
int vo_or_solve_cubic(const double* c4, double* roots)
{
    double a0 = c4[0], a1 = c4[1], a2 = c4[2], a3 = c4[3];
    double x0 = 0, x1 = 0, x2 = 0;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = om_acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * om_cos(t1) - t2;
            x1 = t0 * om_cos(t1 + (2. * OR_PI / 3)) - t2;
            x2 = t0 * om_cos(t1 + (4. * OR_PI / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * om_cbrt_pow(R) - a1 / 3; x1 = om_cbrt_pow(R) - a1 / 3; }
            else { x0 = 2 * om_cbrt_pow(-R) - a1 / 3; x1 = -om_cbrt_pow(-R) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            d = sqrt(-d);
            double e = om_cbrt_pow(d + fabs(R));
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0; roots[1] = x1; roots[2] = x2;
    return n;
}

/* ------------------------------------------------------------------ polynom_solver */
static int solve_deg2(double a, double b, double c, double* x1, double* x2)
{
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { *x1 = -b * inv_2a; *x2 = *x1; return 1; }
    double sq = sqrt(delta);
    *x1 = (-b + sq) * inv_2a;
    *x2 = (-b - sq) * inv_2a;
    return 2;
}

static int solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c, d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { *x0 = *x1 = *x2 = -b_a_3; return 3; }
        /* pow(2R,1/3) of the published routine is NaN for R<0; keep that behaviour in
         * libm mode, det mode uses the real cube root of |2R| with R's sign dropped too */
        *x0 = (g_or_det() ? (2 * R > 0 ? om_cbrt_pow(2 * R) : NAN) : pow(2 * R, 1 / 3.0)) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        double theta = om_acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * om_cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * om_cos((theta + 2 * OR_PI) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * om_cos((theta + 4 * OR_PI) / 3.0) - b_a_3;
        return 3;
    }
    double AD = om_cbrt_pow(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

int vo_or_solve_deg4(double a, double b, double c, double d, double e, double* x)
{
    x[0] = x[1] = x[2] = x[3] = 0;
    if (a == 0) return solve_deg3(b, c, d, e, &x[0], &x[1], &x[2]);
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0 = 0, r1 = 0, r2 = 0;
    int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0) return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;
    int nb = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            double st = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * st;
            E2 = D2 - 4 * st;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb = 2;
        double D_2 = 0.5 * D;
        x[0] = R_2 + D_2 - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2), E_2 = 0.5 * E;
        if (nb == 0) { x[0] = -R_2 + E_2 - b_4; x[1] = x[0] - E; nb = 2; }
        else { x[2] = -R_2 + E_2 - b_4; x[3] = x[2] - E; nb = 4; }
    }
    return nb;
}

/* ------------------------------------------------------------------ cv::solvePoly */
/* coeffs[0..deg] ascending (coeffs[0] constant).  Durand-Kerner from the start points
 * (1+i)^k, up to 1000 sweeps, stops when no root moved.  returns #roots. */
int vo_or_solve_poly(const double* coeffs, int deg, double* re, double* im)
{
    int n = deg;
    for (; n > 1; --n)
        if (fabs(coeffs[n]) > DBL_EPSILON) break;
    double pr = 1, pi = 0;
    for (int i = 0; i < n; ++i) {
        re[i] = pr; im[i] = pi;
        double nr = pr * 1 - pi * 1, ni = pr * 1 + pi * 1;
        pr = nr; pi = ni;
    }
    /* cv::solvePoly's default cap (core.hpp: maxIters = 300), which five-point.cpp runs with [OCV]; the update almost
     * never reaches an exact fixpoint, so all of them run */
    for (int iter = 0; iter < 300; ++iter) {
        double maxdiff = 0;
        for (int i = 0; i < n; ++i) {
            double xr = re[i], xi = im[i];
            double nr = coeffs[n], ni = 0, dr = coeffs[n], di = 0;
            for (int j = 0; j < n; ++j) {
                double tr = nr * xr - ni * xi + coeffs[n - j - 1];
                double ti = nr * xi + ni * xr;
                nr = tr; ni = ti;
                if (j != i) {
                    double er = xr - re[j], ei = xi - im[j];
                    if (er != 0 || ei != 0) {
                        tr = dr * er - di * ei;
                        ti = dr * ei + di * er;
                        dr = tr; di = ti;
                    }
                }
            }
            double t = 1. / (dr * dr + di * di);
            double qr = (nr * dr + ni * di) * t, qi = (-nr * di + ni * dr) * t;
            re[i] = xr - qr; im[i] = xi - qi;
            double ad = sqrt(qr * qr + qi * qi);
            if (ad > maxdiff) maxdiff = ad;
        }
        if (maxdiff <= 0) break;
    }
    for (int i = 0; i < n; ++i)
        if (fabs(im[i]) < 1e-100) im[i] = 0;
    return n;
}
