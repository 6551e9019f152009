// vo_epnp_dev.hip -- the EPnP refit of cv.solvePnPRansac(flags=SOLVEPNP_P3P) (reference src/pnp.py:16-26;
// SURVEY.md Appendix A.5) for a batch of frame pairs on the device: the pose OpenCV returns is EPnP over
// the inliers of the winning P3P model, so the batched pipeline needs it next to the RANSAC results.
//
// One 128-thread workgroup per pair (VO_EPNP_THREADS).  Everything that is O(inliers) -- centroid, covariance, the 12x12
// normal matrix M^T M, the cross-covariance of the absolute orientation, the mean reprojection error
// -- is a block reduction in a fixed tree order (deterministic run to run); the small dense algebra
// in between (3x3 / 12x12 Jacobi SVD, the 6x10 system of the betas, Gauss-Newton) is the code of the
// host refit (vo_epnp_core.h) executed by thread 0.  The sums are associated differently from the
// host's sequential loops, so poses agree with vo_host_epnp to rounding (tests: 1e-6), not bit for bit.
#include "vo_common.h"
#include "vo_epnp_core.h"

namespace {

using namespace vo_epnp;

#ifndef VO_EPNP_THREADS
#define VO_EPNP_THREADS 128             // 2 waves: the kernel is latency-bound and shares its CU with the F chain
#endif
constexpr int kThreads = VO_EPNP_THREADS;
constexpr int kWaves = kThreads / 64;

// sum of v[k] over the workgroup, result in every thread; sh holds kWaves * K doubles
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* sh)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        v[k] = s;
    }
    __syncthreads();                               // sh may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) sh[(threadIdx.x >> 6) * K + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = sh[k];
        for (int w = 1; w < kWaves; ++w) s += sh[w * K + k];
        v[k] = s;
    }
}

// sum over the 16 lanes of a DPP row; every lane ends with the same bits (each step adds the same two
// partial sums in either order).  Data-parallel-primitive moves instead of LDS permutes: the Jacobi
// sweeps below are one long dependency chain and three of these sit in every rotation.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const unsigned long long u = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double sum16(double x)
{
    x += dpp_f64<0xB1>(x);       // quad_perm [1,0,3,2]: lane ^ 1
    x += dpp_f64<0x4E>(x);       // quad_perm [2,3,0,1]: lane ^ 2
    x += dpp_f64<0x141>(x);      // row_half_mirror: the other quad of the half row
    x += dpp_f64<0x140>(x);      // row_mirror: the other half row
    return x;
}

// The 12 x 12 SVD of M^T M on one 16-lane group: lane k keeps column k of the 12 working rows in
// registers (rows = columns of the symmetric input), row pairs are rotated in OpenCV's sweep order,
// the three dot products of a rotation are 16-lane butterflies.  Returns, through nullv (LDS, 4 x 12),
// the left singular vectors of the four smallest singular values, smallest first (V is not needed).
__device__ void null_space_12(const double* mtm /* LDS, 144 */, double* nullv /* LDS, 48 */, int lane16)
{
    const double tol = DBL_EPSILON * 10;
    const int k = lane16;
    double A[12], nrm2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) A[i] = k < 12 ? mtm[12 * k + i] : 0.;      // a[i][k] = s[k][i]
#pragma unroll
    for (int i = 0; i < 12; ++i) nrm2[i] = sum16(A[i] * A[i]);
    // only the pairs (0, j) exist as code: after row i has met its partners the rows rotate up by one
    // (register moves), the rows already done wrap to the end where the shrinking partner count
    // leaves them alone; twelve rotations restore the order (same device as the 7-point solver)
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll 1
        for (int i = 0; i < 12; ++i) {
#pragma unroll
            for (int j = 1; j < 12; ++j) {
                if (j > 11 - i) continue;                                     // uniform
                const double dot = sum16(A[0] * A[j]);
                if (fabs(dot) <= tol * sqrt(nrm2[0] * nrm2[j])) continue;     // uniform over the group
                const double p = 2 * dot, diff = nrm2[0] - nrm2[j], hyp = hypot(p, diff);
                double c, s;
                if (diff < 0) {
                    s = sqrt((hyp - diff) * 0.5 / hyp);
                    c = p / (hyp * s * 2);
                } else {
                    c = sqrt((hyp + diff) / (hyp * 2));
                    s = p / (hyp * c * 2);
                }
                const double x = c * A[0] + s * A[j], y = c * A[j] - s * A[0];
                A[0] = x; A[j] = y;
                nrm2[0] = sum16(x * x); nrm2[j] = sum16(y * y);
                rotated = true;
            }
            const double a0 = A[0], n0 = nrm2[0];
#pragma unroll
            for (int r = 0; r < 11; ++r) { A[r] = A[r + 1]; nrm2[r] = nrm2[r + 1]; }
            A[11] = a0; nrm2[11] = n0;
        }
        if (!rotated) break;
    }
    double sigma[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) sigma[i] = sqrt(sum16(A[i] * A[i]));
    // selection sort (descending, first maximum wins) on the order only; then the four last rows
    int ord[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) ord[i] = i;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        int big = i;
#pragma unroll
        for (int q = i + 1; q < 12; ++q)
            if (sigma[big] < sigma[q]) big = q;
#pragma unroll
        for (int q = i + 1; q < 12; ++q)
            if (big == q) {
                const double t = sigma[i]; sigma[i] = sigma[q]; sigma[q] = t;
                const int o = ord[i]; ord[i] = ord[q]; ord[q] = o;
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ord[11 - r];
        double val = 0.;
#pragma unroll
        for (int i = 0; i < 12; ++i) val = row == i ? A[i] : val;
        const double inv = sigma[11 - r] > DBL_MIN ? 1.0 / sigma[11 - r] : 0.0;
        if (k < 12) nullv[12 * r + k] = val * inv;
    }
}

struct EpnpArgs {
    const float* obj; const float2* img; const uint8_t* mask; const int32_t* n_dev; const int32_t* result;
    double* Rt; int32_t* ok;
    int N;
    double fu, fv, uc, vc;
};

// Two waves per SIMD must fit: left alone the compiler takes all 512 registers of a SIMD (256 + 256
// accumulation registers as spill space) for the single-thread algebra, a workgroup then owns its whole
// CU and every kernel of the other stream waits for it (measured: 0.2 ms per 256-pair step).
__global__ __launch_bounds__(kThreads, 2)
void epnp_batch_kernel(const EpnpArgs a)
{
    __shared__ double sh[kWaves * 78];
    __shared__ double s_cw[4][3], s_inv[9], s_cc[3][4][3], s_pose[3][12], s_flag[4];
    __shared__ double s_mtm[144], s_nullv[4][12];
    __shared__ int s_first;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.n_dev ? min(a.n_dev[b], a.N) : a.N;
    const float* obj = a.obj + (size_t)b * a.N * 3;
    const float2* img = a.img + (size_t)b * a.N;
    const uint8_t* mask = a.mask + (size_t)b * a.N;
    double* Rt = a.Rt + (size_t)b * 12;
    const bool have = !a.result || a.result[4 * b] >= 0;

    auto fail = [&]() {                            // block-uniform exits only
        if (tid == 0) {
            for (int k = 0; k < 12; ++k) Rt[k] = 0.;
            a.ok[b] = 0;
        }
    };

    // Every O(inliers) pass walks the pair's correspondences four at a time per thread: the loads of
    // the four records are issued together (indices clamped, nothing depends on the mask yet), so a
    // pass costs one memory latency per four points instead of one per point.
    struct Rec { float x, y, z, u, v; bool in; };      // f32 as stored: converting at use is exact
    auto for_inliers = [&](auto&& body) {
        for (int base = 0; base < n; base += 4 * kThreads) {
            Rec r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = base + q * kThreads + tid;
                const int ii = i < n ? i : 0;
                const uint8_t mk = mask[ii];
                const float ox = obj[3 * ii], oy = obj[3 * ii + 1], oz = obj[3 * ii + 2];
                const float2 uv = img[ii];
                r[q].x = ox; r[q].y = oy; r[q].z = oz; r[q].u = uv.x; r[q].v = uv.y;
                r[q].in = i < n && mk != 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r[q].in) body(r[q], base + q * kThreads + tid);
        }
    };

    // ---- pass 1: inlier count, the first inlier, centroid sums
    if (tid == 0) s_first = 0x7fffffff;
    __syncthreads();
    double s1[4] = {0., 0., 0., 0.};
    int first = 0x7fffffff;
    for_inliers([&](const Rec& r, int i) {
        s1[0] += 1.; s1[1] += r.x; s1[2] += r.y; s1[3] += r.z;
        first = min(first, i);
    });
    if (first != 0x7fffffff) atomicMin(&s_first, first);
    block_sum(s1, sh);
    const int m = (int)s1[0];
    if (!have || m < 4) { fail(); return; }
    const double inv_m = 1.0 / m;
    const double c0[3] = {s1[1] / m, s1[2] / m, s1[3] / m};   // centroid: control point 0, and mw of the orientation step

    // ---- pass 2: covariance -> principal axes -> control points
    double cv[6] = {0., 0., 0., 0., 0., 0.};       // xx xy xz yy yz zz
    for_inliers([&](const Rec& r, int) {
        const double d0 = r.x - c0[0], d1 = r.y - c0[1], d2 = r.z - c0[2];
        cv[0] += d0 * d0; cv[1] += d0 * d1; cv[2] += d0 * d2; cv[3] += d1 * d1; cv[4] += d1 * d2; cv[5] += d2 * d2;
    });
    block_sum(cv, sh);
    if (tid == 0) {
        const double cov[9] = {cv[0], cv[1], cv[2], cv[1], cv[3], cv[4], cv[2], cv[4], cv[5]};
        double sig[3], u[9];
        sym_svd_u3(cov, sig, u);
        for (int j = 0; j < 3; ++j) s_cw[0][j] = c0[j];
        for (int ax = 0; ax < 3; ++ax) {
            const double k = sqrt(sig[ax] / m);
            for (int j = 0; j < 3; ++j) s_cw[ax + 1][j] = c0[j] + k * u[3 * ax + j];
        }
        double basis[9], inv[9];
        for (int r = 0; r < 3; ++r)
            for (int ax = 0; ax < 3; ++ax) basis[3 * r + ax] = s_cw[ax + 1][r] - s_cw[0][r];
        const bool okb = inv3(basis, inv);
        for (int k = 0; k < 9; ++k) s_inv[k] = okb ? inv[k] : 0.;
        s_flag[0] = okb ? 1. : 0.;
    }
    __syncthreads();
    if (s_flag[0] == 0.) { fail(); return; }

    auto alphas = [&](double x, double y, double z, double (&al)[4]) {   // barycentric coordinates of an object point
        const double d0 = x - s_cw[0][0], d1 = y - s_cw[0][1], d2 = z - s_cw[0][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) al[1 + j] = s_inv[3 * j] * d0 + s_inv[3 * j + 1] * d1 + s_inv[3 * j + 2] * d2;
        al[0] = 1.0 - al[1] - al[2] - al[3];
    };

    // ---- M^T M (upper triangle, 78 sums)
    double acc[78];
#pragma unroll
    for (int k = 0; k < 78; ++k) acc[k] = 0.;
    for_inliers([&](const Rec& r, int) {
        double al[4];
        alphas(r.x, r.y, r.z, al);
        const double du = a.uc - r.u, dv = a.vc - r.v;
        double r1[12], r2[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r1[3 * k] = al[k] * a.fu; r1[3 * k + 1] = 0; r1[3 * k + 2] = al[k] * du;
            r2[3 * k] = 0; r2[3 * k + 1] = al[k] * a.fv; r2[3 * k + 2] = al[k] * dv;
        }
        int e = 0;
#pragma unroll
        for (int rr = 0; rr < 12; ++rr)
#pragma unroll
            for (int c = rr; c < 12; ++c, ++e) {
                acc[e] += r1[rr] * r1[c];
                acc[e] += r2[rr] * r2[c];
            }
    });
    block_sum(acc, sh);
    __syncthreads();                               // everyone is done reading sh
    if (tid == 0) {
        int e = 0;
        for (int r = 0; r < 12; ++r)
            for (int c = r; c < 12; ++c, ++e) s_mtm[12 * r + c] = s_mtm[12 * c + r] = acc[e];
    }
    __syncthreads();
    if (tid < 16) null_space_12(s_mtm, &s_nullv[0][0], tid);
    __syncthreads();
    if (tid == 0) {
        double nullv[4][12], cw[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 12; ++k) nullv[i][k] = s_nullv[i][k];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) cw[i][k] = s_cw[i][k];
        double L[6][10], rho[6];
        build_L(nullv, cw, L, rho);
#pragma unroll
        for (int N = 1; N <= 3; ++N) {
            double beta[4];
            beta_start(L, rho, N, beta);
            refine_betas(L, rho, beta);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double s = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s += beta[i] * nullv[i][3 * j + k];   // same order as the host's += loop
                    s_cc[N - 1][j][k] = s;
                }
        }
    }
    __syncthreads();

    // ---- pass 4: the sums the absolute orientation needs are linear in the barycentric coordinates, so
    // one pass serves all three candidates: abar_j = sum_i a_ij, B_j = sum_i a_ij (pw_i - mw),
    // D = sum_i (pw_i - mw).  With pc_i = sgn sum_j a_ij cc_j:
    //   mc = sgn sum_j abar_j cc_j / m,   H = sum_i (pc_i - mc)(pw_i - mw)^T = sgn sum_j cc_j B_j^T - mc D^T
    double lin[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) lin[k] = 0.;
    for_inliers([&](const Rec& r, int) {
        double al[4];
        alphas(r.x, r.y, r.z, al);
        const double w[3] = {r.x - c0[0], r.y - c0[1], r.z - c0[2]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lin[j] += al[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) lin[4 + 3 * j + k] += al[j] * w[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) lin[16 + k] += w[k];
    });
    block_sum(lin, sh);
    if (tid == 0) {
        const int i0 = s_first;
        double al0[4];
        alphas((double)obj[3 * i0], (double)obj[3 * i0 + 1], (double)obj[3 * i0 + 2], al0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double z0 = 0.;
#pragma unroll
            for (int j = 0; j < 4; ++j) z0 += al0[j] * s_cc[c][j][2];
            const double sgn = z0 < 0 ? -1.0 : 1.0;        // the host flips every point when the first has z < 0
            double mc[3], H[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double s = 0.;
#pragma unroll
                for (int j = 0; j < 4; ++j) s += lin[j] * s_cc[c][j][k];
                mc[k] = sgn * s / m;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double s = 0.;
#pragma unroll
                    for (int j = 0; j < 4; ++j) s += s_cc[c][j][r] * lin[4 + 3 * j + k];
                    H[3 * r + k] = sgn * s - mc[r] * lin[16 + k];
                }
            double R[9], t[3];
            pose_from_h(H, mc, c0, R, t);
#pragma unroll
            for (int k = 0; k < 9; ++k) s_pose[c][k] = R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) s_pose[c][9 + k] = t[k];
        }
    }
    __syncthreads();

    // ---- pass 5: mean reprojection error of the three candidates; the lowest wins (first on ties)
    double err[3] = {0., 0., 0.};
    for_inliers([&](const Rec& r, int) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double* R = s_pose[c];
            const double* t = s_pose[c] + 9;
            const double X = R[0] * r.x + R[1] * r.y + R[2] * r.z + t[0];
            const double Y = R[3] * r.x + R[4] * r.y + R[5] * r.z + t[1];
            const double iz = 1.0 / (R[6] * r.x + R[7] * r.y + R[8] * r.z + t[2]);
            const double du = r.u - (a.uc + a.fu * X * iz), dv = r.v - (a.vc + a.fv * Y * iz);
            err[c] += sqrt(du * du + dv * dv);
        }
    });
    block_sum(err, sh);
    if (tid == 0) {
        int best = 0;
        for (int c = 1; c < 3; ++c)
            if (err[c] * inv_m < err[best] * inv_m) best = c;
        bool fin = true;
        for (int k = 0; k < 12; ++k) fin = fin && isfinite(s_pose[best][k]);
        for (int k = 0; k < 12; ++k) Rt[k] = fin ? s_pose[best][k] : 0.;
        a.ok[b] = fin ? 1 : 0;
    }
}

}  // namespace

extern "C" int vo_epnp_batch(const float* obj, const float* img, const uint8_t* mask, const int32_t* n_dev,
                             const int32_t* result, const double* K4_host, int N, int batch, double* Rt, int32_t* ok,
                             void* stream)
{
    VO_CHECK_ARG(N >= 0 && batch >= 0, VO_EINVAL, "vo_epnp_batch: negative size");
    if (batch == 0) return VO_OK;
    VO_CHECK_ARG(K4_host && Rt && ok, VO_EINVAL, "vo_epnp_batch: null pointer");
    VO_CHECK_ARG(N == 0 || (obj && img && mask), VO_EINVAL, "vo_epnp_batch: null point buffers");
    VO_CHECK_ARG(((uintptr_t)img & 7) == 0, VO_EINVAL, "vo_epnp_batch: img must be 8-byte aligned");
    EpnpArgs a;
    a.obj = obj; a.img = (const float2*)img; a.mask = mask; a.n_dev = n_dev; a.result = result; a.Rt = Rt; a.ok = ok;
    a.N = N; a.fu = K4_host[0]; a.fv = K4_host[1]; a.uc = K4_host[2]; a.vc = K4_host[3];
    hipLaunchKernelGGL(epnp_batch_kernel, dim3(batch), dim3(kThreads), 0, (hipStream_t)stream, a);
    VO_LAUNCH_CHECK();
    return VO_OK;
}
