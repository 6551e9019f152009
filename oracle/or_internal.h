/* or_internal.h -- CPU ORACLE internals (test infrastructure, NOT product code). */
#ifndef OR_INTERNAL_H
#define OR_INTERNAL_H
#include "vo_oracle.h"

#define OR_PI 3.1415926535897932384626433832795

static inline int g_or_det(void) { return vo_or_get_det_math(); }

double om_acos(double x);
double om_cos(double x);
double om_sin(double x);
double om_cbrt_pow(double x);          /* pow(x, 1./3) for x >= 0 */
double om_hypot(double a, double b);

#endif
