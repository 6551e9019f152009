// vo_detmath.h -- elementary functions for the device solvers, built from IEEE + - * / sqrt,
// floor, frexp and ldexp only, so that every value is reproducible bit for bit by a scalar CPU
// evaluation of the same formulas (the device maths library's acos/cos/sin/cbrt are accurate but
// not bit-identical to any CPU libm).  Kernels: the minimax polynomials / rational of fdlibm
// (Sun Microsystems, freely distributable) on their reduced ranges, evaluated in Horner form;
// cube root by exponent split and six Newton steps.  Accuracy ~1-2 ulp; inputs are limited to the
// ranges the solvers produce (angles in [0, ~100), |x| <= 1 for acos, x >= 0 for cbrt).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace vodm {

__device__ __forceinline__ double k_sin(double x)
{
    const double z = x * x;
    double p = 1.58969099521155010221e-10;
    p = p * z + -2.50507602534068634195e-08;
    p = p * z + 2.75573137070700676789e-06;
    p = p * z + -1.98412698298579493134e-04;
    p = p * z + 8.33333333332248946124e-03;
    p = p * z + -1.66666666666666324348e-01;
    return x + x * (z * p);
}

__device__ __forceinline__ double k_cos(double x)
{
    const double z = x * x;
    double p = -1.13596475577881948265e-11;
    p = p * z + 2.08757232129817482790e-09;
    p = p * z + -2.75573143513906633035e-07;
    p = p * z + 2.48015872894767294178e-05;
    p = p * z + -1.38888888888741095749e-03;
    p = p * z + 4.16666666666666019037e-02;
    return (1.0 - 0.5 * z) + z * (z * p);
}

__device__ __forceinline__ void sincos(double x, double* s, double* c)
{
    const double kf = floor(x * 6.36619772367581382433e-01 + 0.5);
    const int k = (int)kf;
    const double r = (x - kf * 1.57079632673412561417e+00) - kf * 6.07710050650619224932e-11;
    const double sr = k_sin(r), cr = k_cos(r);
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

__device__ __forceinline__ double cos_(double x) { double s, c; sincos(x, &s, &c); return c; }
__device__ __forceinline__ double sin_(double x) { double s, c; sincos(x, &s, &c); return s; }

__device__ __forceinline__ double acos_(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    const double ax = fabs(x);
    if (ax >= 1.0) return x > 0 ? 0.0 : pi;
    const double z = ax < 0.5 ? x * x : (1.0 - ax) * 0.5;
    double p = 3.47933107596021167570e-05;
    p = p * z + 7.91534994289814532176e-04;
    p = p * z + -4.00555345006794114027e-02;
    p = p * z + 2.01212532134862925881e-01;
    p = p * z + -3.25565818622400915405e-01;
    p = p * z + 1.66666666666666657415e-01;
    p = p * z;
    double q = 7.70381505559019352791e-02;
    q = q * z + -6.88283971605453293030e-01;
    q = q * z + 2.02094576023350569471e+00;
    q = q * z + -2.40339491173441421878e+00;
    q = q * z + 1.0;
    const double r = p / q;
    if (ax < 0.5) return pio2_hi - (x - (pio2_lo - x * r));
    const double s = sqrt(z);
    const double w = s + s * r;
    return x > 0 ? 2.0 * w : pi - 2.0 * w;
}

__device__ __forceinline__ double cbrt_(double x)
{
    if (!(x > 0.0)) return 0.0;
    int e;
    double m = frexp(x, &e);
    int r = e % 3;
    if (r < 0) r += 3;
    if (r) { m = ldexp(m, r - 3); e += 3 - r; }
    double y = 0.4285714285714286 + 0.5714285714285714 * m;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const double y2 = y * y;
        y = y - (y2 * y - m) / (3.0 * y2);
    }
    return ldexp(y, e / 3);
}

}  // namespace vodm
