// VD-CMA, device-resident generation: everything between two looks of the host at the 128-byte state.
//
// Reference code replaced (paths relative to the reference checkout), on top of the kernels of sx_cmaes.hip
// (candidates :236-248, the O(mu n) moment sums :289-295, :317, :331-339) and of sx_cma_loop.hip (ranking, history):
//   stochopy/optimize/vdcma/_vdcma.py:241-247  mean-shift injection: dy = |z| / sqrt(mnorm) * dx
//   stochopy/optimize/vdcma/_vdcma.py:292-295  dx, xold, xmean
//   stochopy/optimize/vdcma/_vdcma.py:298-306  step size from the rank gap of the injected pair
//   stochopy/optimize/vdcma/_vdcma.py:309-328  evolution path, alpha / beta / b, avec, invavnn
//   stochopy/optimize/vdcma/_vdcma.py:331-378  moments of the path, natural gradient (:447-460), update of v and d
//   stochopy/optimize/cmaes/_cmaes.py:360-434  converge as vdcma calls it (no B, D: rules -2 and -4 drop out)
// The model update is O(n): one workgroup of 512 threads holds the vectors in registers (n <= 4096), with a workgroup
// reduction wherever the reference takes a dot product, a norm, a max or a min; the scalars in between are computed by
// every thread from the reduced values.  Expressions keep the reference's association order (no FMA contraction in
// this build).
#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace sx {
int vd_sample_launch(const double *Z, int64_t P, int n, const double *dvec, const double *vn, const double *xmean,
                     const double *dy, double *ary, double *arx, const sx_cma_state *st, void *stream, int64_t row0 = 0);
int vd_moments_launch(const double *arx, const double *ary, const int64_t *idx, const double *w, int mu, int n,
                      const double *dvec, const double *vn, double norm_v2, const sx_cma_state *state, double *ws,
                      double *out, void *stream);
int cma_rank_launch(const double *fit, int64_t P, int64_t *order, sx_cma_state *state, double *besthist, int64_t gen,
                    void *stream);
int cma_history_launch(const sx_cma_args &h, int64_t gen, void *stream);
int cma_penalize_launch(const sx_cma_args &h, int64_t gen, const double *dvec, const double *vvec, void *stream);
}  // namespace sx

namespace {

constexpr int kVdThreads = 512;
constexpr int kVdWaves = kVdThreads / 64;
constexpr int kVdPer = 8;  // elements per thread: n <= 4096, everything in registers (227 VGPRs; 1024 threads x 4 spill 154)

// K values at once (kind[q]: 0 sum, 1 max, 2 min): one pair of barriers for all of them; every thread gets the results
template <int K, int NW = kVdWaves>
__device__ void reduce_many(double (&v)[K], const int (&kind)[K], double (*red)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(v[q], off, kWave);
            v[q] = kind[q] == 0 ? v[q] + o : (kind[q] == 1 ? fmax(v[q], o) : fmin(v[q], o));
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) red[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < K; ++q) {
        double r = red[0][q];
        for (int wv = 1; wv < NW; ++wv) {
            const double o = red[wv][q];
            r = kind[q] == 0 ? r + o : (kind[q] == 1 ? fmax(r, o) : fmin(r, o));
        }
        v[q] = r;
    }
}

__device__ double reduce_sum(double x, double (*red)[1]) {
    double v[1] = {x};
    const int k[1] = {0};
    reduce_many<1>(v, k, red);
    return v[0];
}

// the mean-shift injection (:241-247): dy = |z| / sqrt(mnorm) * dx with mnorm = |dx/d|^2 - (dx/d . v)^2 / (1 + |v|^2)
__global__ __launch_bounds__(kVdThreads) void vd_inject_kernel(const sx_vd_args a) {
    __shared__ double red3[kVdWaves][3];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done || state->reserved[3] == 0.0) return;
    const int n = a.n, tid = threadIdx.x;
    const double nv2 = state->reserved[1];
    double dxe[kVdPer];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        dxe[u] = 0.0;
        if (e < n) {
            dxe[u] = a.dx[e];
            const double ddx = dxe[u] / a.dvec[e], z = a.zinj[e];
            s1 += ddx * ddx;
            s2 += ddx * a.vvec[e];
            s3 += z * z;
        }
    }
    double v3[3] = {s1, s2, s3};
    const int k3[3] = {0, 0, 0};
    reduce_many<3>(v3, k3, red3);
    const double mnorm = v3[0] - v3[1] * v3[1] / (1.0 + nv2);
    const double fac = sqrt(v3[2]) / sqrt(mnorm);
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        if (e < n) a.dy[e] = fac * dxe[u];
    }
}

__global__ __launch_bounds__(kVdThreads) void vd_update_kernel(const sx_vd_args a, int64_t gen) {
    __shared__ double red1[kVdWaves][1];
    __shared__ double red2[kVdWaves][2];
    __shared__ double red4[kVdWaves][4];
    __shared__ double red10[kVdWaves][10];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x;
    const double sigma0 = state->sigma, ps0 = state->reserved[0], nv2 = state->reserved[1], nv = state->reserved[2];
    const bool inject = state->reserved[3] != 0.0;
    const double fbest = state->fbest;
    const double *wx = a.mout, *wy = a.mout + n, *pmu = a.mout + 2 * (int64_t)n, *qmu = a.mout + 3 * (int64_t)n;

    // ---- where the injected pair (rows 0 and 1) ended up in the ranking (:299-300) ----
    double pos0 = 0.0, pos1 = 0.0;
    if (inject) {
        for (int64_t k = tid; k < a.P; k += kVdThreads) {
            const int64_t r = a.order[k];
            if (r == 0) pos0 = (double)k;
            if (r == 1) pos1 = (double)k;
        }
    }
    // ---- mean shift (:292-294).  Registers are the budget: vectors are kept only while they are
    // needed, and cheap ones (vn^2, invavnn, the standard deviations) are formed again where they are used ----
    double dv[kVdPer], vv[kVdPer], v1[kVdPer];
    double dx2 = 0.0, vmax = -__builtin_inf();
    {
        double xm0[kVdPer], wxe[kVdPer];
#pragma unroll
        for (int u = 0; u < kVdPer; ++u) {
            const int e = tid + u * kVdThreads;
            const bool in = e < n;
            xm0[u] = in ? a.xmean[e] : 0.0;
            wxe[u] = in ? wx[e] : 0.0;
            dv[u] = in ? a.dvec[e] : 1.0;
            vv[u] = in ? a.vvec[e] : 0.0;
            v1[u] = in ? a.vn[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kVdPer; ++u) {
            const int e = tid + u * kVdThreads;
            if (e < n) {
                const double dxe = wxe[u] - a.wsum * xm0[u];
                a.dx[e] = dxe;
                a.xold[e] = xm0[u];
                a.xmean[e] = xm0[u] + dxe;
                dx2 += dxe * dxe;
                vmax = fmax(vmax, v1[u] * v1[u]);
            }
        }
    }
    double pc0[kVdPer], wye[kVdPer];
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {  // requested before the reduction, used after it
        const int e = tid + u * kVdThreads;
        pc0[u] = e < n ? a.pc[e] : 0.0;
        wye[u] = e < n ? wy[e] : 0.0;
    }
    {
        double v4[4] = {pos0, pos1, dx2, vmax};
        const int k4[4] = {0, 0, 0, 1};
        reduce_many<4>(v4, k4, red4);
        pos0 = v4[0], pos1 = v4[1], dx2 = v4[2], vmax = v4[3];
    }
    // ---- step size from the rank gap (:298-306) ----
    double ps = ps0, sigma = sigma0;
    bool cond = true;
    if (inject) {
        const double gap = (pos1 - pos0) / ((double)a.P - 1.0);
        ps = ps0 + a.cs * (gap - ps0);
        sigma = sigma0 * exp(ps / a.ds);
        cond = ps < 0.5;
    }
    // ---- model constants (:317-328) ----
    const double gamma = 1.0 / sqrt(1.0 + nv2);
    double alpha = sqrt(nv2 * nv2 + (1.0 + nv2) / vmax * (2.0 - gamma)) / (2.0 + nv2);
    double beta = 0.0;
    if (alpha < 1.0) {
        const double t2 = 1.0 + 2.0 / nv2;
        beta = (4.0 - (2.0 - gamma) / vmax) / (t2 * t2);
    } else {
        alpha = 1.0;
    }
    const double bsca = 2.0 * (alpha * alpha) - beta;
    // ---- evolution path (:309-314), y = pc / d, t = y . vn; avec (invavnn = vn^2 / avec) ----
    const double cpc = sqrt(a.cc * (2.0 - a.cc) * a.mueff);
    double y[kVdPer], avec[kVdPer];
    double t = 0.0, svi = 0.0;
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        y[u] = 0.0, avec[u] = 1.0;
        if (e < n) {
            double pce = pc0[u] * (1.0 - a.cc);
            if (cond) pce = pce + cpc * wye[u];
            a.pc[e] = pce;
            pc0[u] = pce;
            y[u] = pce / dv[u];
            t += y[u] * v1[u];
            const double vnn = v1[u] * v1[u];
            avec[u] = 2.0 - (bsca + 2.0 * (alpha * alpha)) * vnn;
            svi += vnn * (vnn / avec[u]);
        }
    }
    double p[kVdPer], q[kVdPer];
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {  // requested before the reduction, used after it
        const int e = tid + u * kVdThreads;
        p[u] = (e < n && a.cmu != 0.0) ? a.cmu * pmu[e] : 0.0;
        q[u] = (e < n && a.cmu != 0.0) ? a.cmu * qmu[e] : 0.0;
    }
    {
        double v2[2] = {t, svi};
        const int k2[2] = {0, 0};
        reduce_many<2>(v2, k2, red2);
        t = v2[0], svi = v2[1];
    }
    // ---- moments of the path (:340-345, :428-444), p and q (:348-352), vn . q ----
    const double shrink = nv2 / (1.0 + nv2);
    double vq = 0.0;
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        if (e < n) {
            if (cond && a.c1 != 0.0) {
                const double p_one = (y[u] * y[u] - shrink * ((t * y[u]) * v1[u])) - 1.0;
                const double q_one = t * y[u] - (0.5 * ((t * t + 1.0) + nv2)) * v1[u];
                p[u] = p[u] + a.c1 * p_one;
                q[u] = q[u] + a.c1 * q_one;
            }
            vq += v1[u] * q[u];
        }
    }
    vq = reduce_sum(vq, red1);
    const bool learn = a.cmu + a.c1 > 0.0;
    // ---- natural gradient (:447-460) ----
    double ri = 0.0;
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {  // r overwrites p
        const int e = tid + u * kVdThreads;
        if (e < n) {
            const double vnn = v1[u] * v1[u];
            p[u] = p[u] - alpha / (1.0 + nv2) * (((2.0 + nv2) * q[u]) * v1[u] - (nv2 * vq) * vnn);
            ri += p[u] * (vnn / avec[u]);
        } else {
            p[u] = 0.0;
        }
    }
    ri = reduce_sum(ri, red1);
    double svn = 0.0;
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {  // s overwrites r
        const int e = tid + u * kVdThreads;
        if (e < n) {
            const double vnn = v1[u] * v1[u];
            p[u] = p[u] / avec[u] - bsca * ri / (1.0 + bsca * svi) * (vnn / avec[u]);
            svn += p[u] * vnn;
        }
    }
    svn = reduce_sum(svn, red1);
    double g2 = 0.0, mind = __builtin_inf();
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {  // ngv overwrites q, ngd overwrites s
        const int e = tid + u * kVdThreads;
        if (e < n && learn) {
            q[u] = q[u] / nv - alpha / nv * ((2.0 + nv2) * (v1[u] * p[u]) - svn * v1[u]);
            p[u] = dv[u] * p[u];
            g2 += q[u] * q[u];
            mind = fmin(mind, dv[u] / fabs(p[u]));
        } else {
            q[u] = 0.0, p[u] = 0.0;
        }
    }
    {
        double v2[2] = {g2, mind};
        const int k2[2] = {0, 2};
        reduce_many<2>(v2, k2, red2);
        g2 = v2[0], mind = v2[1];
    }
    double up = 1.0;
    if (learn) {
        up = fmin(1.0, 0.7 * nv / sqrt(g2));
        up = fmin(up, 0.7 * mind);
    }
    // ---- update of v and d (:371-378); the stopping rules' per-dimension counts on the model the candidates were
    // drawn with and the NEW sigma, pc, mean; the best-fitness histories (window [gen - ilim, gen] of the
    // zero-initialised array, and the whole array joined with this generation's fitness values) ----
    double nv2n = 0.0, any3 = 0.0, any6 = 0.0, fail8 = 0.0, nan_sd = 0.0, sdmax = -__builtin_inf();
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        if (e < n) {
            const double sd = sqrt((dv[u] * (1.0 + vv[u] * vv[u])) * dv[u]);  // sqrt of diag D (I + v v^T) D (:249-254)
            vv[u] = vv[u] + up * q[u];
            a.vvec[e] = vv[u];
            a.dvec[e] = dv[u] + up * p[u];
            nv2n += vv[u] * vv[u];
            if (0.2 * sigma * sd < 1.0e-10) any3 += 1.0;
            if (sigma * sd > 1.0e3 * a.insigma) any6 += 1.0;
            if (sd != sd) nan_sd += 1.0;
            sdmax = fmax(sdmax, sd);
            if (!(sigma * fabs(pc0[u]) < 1.0e-11 * a.insigma)) fail8 += 1.0;
        }
    }
    double wmax = -__builtin_inf(), wmin = __builtin_inf(), jmax = -__builtin_inf(), jmin = __builtin_inf();
    if (gen >= a.ilim) {
        const int64_t hi = gen + 1 < a.maxiter ? gen + 1 : a.maxiter;
        for (int64_t k = gen - a.ilim + tid; k < hi; k += kVdThreads) {
            const double v = a.besthist[k];
            wmax = fmax(wmax, v), wmin = fmin(wmin, v);
        }
    }
    for (int64_t k = tid; k < a.maxiter; k += kVdThreads) {
        const double v = a.besthist[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    for (int64_t k = tid; k < a.P; k += kVdThreads) {
        const double v = a.fit[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    {
        double v10[10] = {nv2n, any3, any6, fail8, nan_sd, sdmax, wmax, jmax, wmin, jmin};
        const int k10[10] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2};
        reduce_many<10>(v10, k10, red10);
        nv2n = v10[0], any3 = v10[1], any6 = v10[2], fail8 = v10[3], nan_sd = v10[4], sdmax = v10[5], wmax = v10[6];
        jmax = v10[7], wmin = v10[8], jmin = v10[9];
    }
    const double nvn = sqrt(nv2n);
#pragma unroll
    for (int u = 0; u < kVdPer; ++u) {
        const int e = tid + u * kVdThreads;
        if (e < n) a.vn[e] = vv[u] / nvn;
    }
    int status = SX_STATUS_NONE;
    if (gen >= a.maxiter)
        status = -1;
    else if (sqrt(dx2) <= a.xtol && fbest < a.ftol)
        status = 0;
    else if (fbest <= a.ftol)
        status = 1;
    else if (any3 > 0.0)
        status = -3;
    else if (gen >= a.ilim && wmax - wmin < 1.0e-10)
        status = -5;
    else if (any6 > 0.0)
        status = -6;
    else if (gen > 2 && jmax - jmin < 1.0e-12)
        status = -7;
    else if (fail8 == 0.0 && nan_sd == 0.0 && sigma * sdmax < 1.0e-11 * a.insigma)
        status = -8;
    if (status != SX_STATUS_NONE) {  // the caller's result: best candidate of THIS generation, un-standardised
        const double *row = a.arx + state->best_row * (int64_t)n;
        for (int e = tid; e < n; e += kVdThreads) {
            double x = row[e];
            if (a.pen_ws != nullptr) x = fmin(fmax(x, -1.0), 1.0);  // Penalize: the clipped point is what the caller sees
            a.xbest[e] = x * a.xstd[e] + a.xm[e];
        }
    }
    __syncthreads();
    if (tid == 0) {
        state->sigma = sigma;
        state->reserved[0] = ps;
        state->reserved[1] = nv2n;
        state->reserved[2] = nvn;
        state->reserved[3] = 1.0;  // from the second generation on the injection is on (:304-305)
        state->reserved[4] = sqrt(1.0 + nv2n) - 1.0;
        state->it = gen;
        state->nfev = gen * a.P;
        if (status != SX_STATUS_NONE) {
            state->status = status;
            state->stop_it = gen;
            __threadfence();
            state->done = 1;
        }
    }
}

// ---------------------------------------------------------------------------
// The same two kernels for wide models (n > 4096, where VD-CMA matters most: vdcma/_vdcma.py:144-458 is the O(n) companion
// of CMA-ES): the vectors no longer fit a workgroup's registers, so every phase walks them in memory (a few n-vectors, L2
// resident) with 1 024 threads; p / q (later r, s, the natural-gradient steps) wait in two scratch vectors -- the partial
// sums of sx_vdcma_moments, free once its finish kernel has run -- and y, avec, vn^2 are formed again where they are used.
// Same expressions, same association; only the ORDER of the reductions' additions differs from the register kernel
// (both are compared with the oracle to rounding, tests/test_gpu_wide.py).
// ---------------------------------------------------------------------------
constexpr int kVdWideThreads = 1024;
constexpr int kVdWideWaves = kVdWideThreads / 64;

__global__ __launch_bounds__(kVdWideThreads) void vd_inject_wide_kernel(const sx_vd_args a) {
    __shared__ double red3[kVdWideWaves][3];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done || state->reserved[3] == 0.0) return;
    const int n = a.n, tid = threadIdx.x;
    const double nv2 = state->reserved[1];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
    for (int e = tid; e < n; e += kVdWideThreads) {
        const double ddx = a.dx[e] / a.dvec[e], z = a.zinj[e];
        s1 += ddx * ddx;
        s2 += ddx * a.vvec[e];
        s3 += z * z;
    }
    double v3[3] = {s1, s2, s3};
    const int k3[3] = {0, 0, 0};
    reduce_many<3, kVdWideWaves>(v3, k3, red3);
    const double mnorm = v3[0] - v3[1] * v3[1] / (1.0 + nv2);
    const double fac = sqrt(v3[2]) / sqrt(mnorm);
#pragma unroll 4
    for (int e = tid; e < n; e += kVdWideThreads) a.dy[e] = fac * a.dx[e];
}

__global__ __launch_bounds__(kVdWideThreads) void vd_update_wide_kernel(const sx_vd_args a, int64_t gen,
                                                                        double *__restrict__ sp, double *__restrict__ sq) {
    constexpr int T = kVdWideThreads;
    __shared__ double red1[kVdWideWaves][1];
    __shared__ double red2[kVdWideWaves][2];
    __shared__ double red4[kVdWideWaves][4];
    __shared__ double red10[kVdWideWaves][10];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x;
    const double sigma0 = state->sigma, ps0 = state->reserved[0], nv2 = state->reserved[1], nv = state->reserved[2];
    const bool inject = state->reserved[3] != 0.0;
    const double fbest = state->fbest;
    const double *wx = a.mout, *wy = a.mout + n, *pmu = a.mout + 2 * (int64_t)n, *qmu = a.mout + 3 * (int64_t)n;
    auto rsum = [&](double x) {
        double v[1] = {x};
        const int k[1] = {0};
        reduce_many<1, kVdWideWaves>(v, k, red1);
        return v[0];
    };
    // ---- the injected pair in the ranking (:299-300), mean shift (:292-294) ----
    double pos0 = 0.0, pos1 = 0.0;
    if (inject) {
        for (int64_t k = tid; k < a.P; k += T) {
            const int64_t r = a.order[k];
            if (r == 0) pos0 = (double)k;
            if (r == 1) pos1 = (double)k;
        }
    }
    double dx2 = 0.0, vmax = -__builtin_inf();
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        const double xm0 = a.xmean[e], v1 = a.vn[e];
        const double dxe = wx[e] - a.wsum * xm0;
        a.dx[e] = dxe;
        a.xold[e] = xm0;
        a.xmean[e] = xm0 + dxe;
        dx2 += dxe * dxe;
        vmax = fmax(vmax, v1 * v1);
    }
    {
        double v4[4] = {pos0, pos1, dx2, vmax};
        const int k4[4] = {0, 0, 0, 1};
        reduce_many<4, kVdWideWaves>(v4, k4, red4);
        pos0 = v4[0], pos1 = v4[1], dx2 = v4[2], vmax = v4[3];
    }
    // ---- step size from the rank gap (:298-306) ----
    double ps = ps0, sigma = sigma0;
    bool cond = true;
    if (inject) {
        const double gap = (pos1 - pos0) / ((double)a.P - 1.0);
        ps = ps0 + a.cs * (gap - ps0);
        sigma = sigma0 * exp(ps / a.ds);
        cond = ps < 0.5;
    }
    // ---- model constants (:317-328) ----
    const double gamma = 1.0 / sqrt(1.0 + nv2);
    double alpha = sqrt(nv2 * nv2 + (1.0 + nv2) / vmax * (2.0 - gamma)) / (2.0 + nv2);
    double beta = 0.0;
    if (alpha < 1.0) {
        const double t2 = 1.0 + 2.0 / nv2;
        beta = (4.0 - (2.0 - gamma) / vmax) / (t2 * t2);
    } else {
        alpha = 1.0;
    }
    const double bsca = 2.0 * (alpha * alpha) - beta;
    auto avec_of = [&](double vnn) { return 2.0 - (bsca + 2.0 * (alpha * alpha)) * vnn; };
    // ---- evolution path (:309-314), y = pc / d, t = y . vn; avec (invavnn = vn^2 / avec) ----
    const double cpc = sqrt(a.cc * (2.0 - a.cc) * a.mueff);
    double t = 0.0, svi = 0.0;
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        double pce = a.pc[e] * (1.0 - a.cc);
        if (cond) pce = pce + cpc * wy[e];
        a.pc[e] = pce;
        const double v1 = a.vn[e], y = pce / a.dvec[e];
        t += y * v1;
        const double vnn = v1 * v1;
        svi += vnn * (vnn / avec_of(vnn));
    }
    {
        double v2[2] = {t, svi};
        const int k2[2] = {0, 0};
        reduce_many<2, kVdWideWaves>(v2, k2, red2);
        t = v2[0], svi = v2[1];
    }
    // ---- moments of the path (:340-345, :428-444), p and q (:348-352), vn . q ----
    const double shrink = nv2 / (1.0 + nv2);
    double vq = 0.0;
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        const double v1 = a.vn[e], y = a.pc[e] / a.dvec[e];
        double p = a.cmu != 0.0 ? a.cmu * pmu[e] : 0.0;
        double q = a.cmu != 0.0 ? a.cmu * qmu[e] : 0.0;
        if (cond && a.c1 != 0.0) {
            const double p_one = (y * y - shrink * ((t * y) * v1)) - 1.0;
            const double q_one = t * y - (0.5 * ((t * t + 1.0) + nv2)) * v1;
            p = p + a.c1 * p_one;
            q = q + a.c1 * q_one;
        }
        sp[e] = p, sq[e] = q;
        vq += v1 * q;
    }
    vq = rsum(vq);
    const bool learn = a.cmu + a.c1 > 0.0;
    // ---- natural gradient (:447-460): r overwrites p, then s overwrites r ----
    double ri = 0.0;
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        const double v1 = a.vn[e], vnn = v1 * v1;
        const double r = sp[e] - alpha / (1.0 + nv2) * (((2.0 + nv2) * sq[e]) * v1 - (nv2 * vq) * vnn);
        sp[e] = r;
        ri += r * (vnn / avec_of(vnn));
    }
    ri = rsum(ri);
    double svn = 0.0;
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        const double v1 = a.vn[e], vnn = v1 * v1, av = avec_of(vnn);
        const double sv = sp[e] / av - bsca * ri / (1.0 + bsca * svi) * (vnn / av);
        sp[e] = sv;
        svn += sv * vnn;
    }
    svn = rsum(svn);
    double g2 = 0.0, mind = __builtin_inf();
#pragma unroll 4
    for (int e = tid; e < n; e += T) {  // ngv overwrites q, ngd overwrites s
        double qq = 0.0, pp = 0.0;
        if (learn) {
            const double v1 = a.vn[e], dv = a.dvec[e], sv = sp[e];
            qq = sq[e] / nv - alpha / nv * ((2.0 + nv2) * (v1 * sv) - svn * v1);
            pp = dv * sv;
            g2 += qq * qq;
            mind = fmin(mind, dv / fabs(pp));
        }
        sq[e] = qq, sp[e] = pp;
    }
    {
        double v2[2] = {g2, mind};
        const int k2[2] = {0, 2};
        reduce_many<2, kVdWideWaves>(v2, k2, red2);
        g2 = v2[0], mind = v2[1];
    }
    double up = 1.0;
    if (learn) {
        up = fmin(1.0, 0.7 * nv / sqrt(g2));
        up = fmin(up, 0.7 * mind);
    }
    // ---- update of v and d (:371-378); the stopping rules' per-dimension counts; the best-fitness histories ----
    double nv2n = 0.0, any3 = 0.0, any6 = 0.0, fail8 = 0.0, nan_sd = 0.0, sdmax = -__builtin_inf();
#pragma unroll 4
    for (int e = tid; e < n; e += T) {
        const double dv = a.dvec[e], vv0 = a.vvec[e];
        const double sd = sqrt((dv * (1.0 + vv0 * vv0)) * dv);  // sqrt of diag D (I + v v^T) D (:249-254)
        const double vv = vv0 + up * sq[e];
        a.vvec[e] = vv;
        a.dvec[e] = dv + up * sp[e];
        nv2n += vv * vv;
        if (0.2 * sigma * sd < 1.0e-10) any3 += 1.0;
        if (sigma * sd > 1.0e3 * a.insigma) any6 += 1.0;
        if (sd != sd) nan_sd += 1.0;
        sdmax = fmax(sdmax, sd);
        if (!(sigma * fabs(a.pc[e]) < 1.0e-11 * a.insigma)) fail8 += 1.0;
    }
    double wmax = -__builtin_inf(), wmin = __builtin_inf(), jmax = -__builtin_inf(), jmin = __builtin_inf();
    if (gen >= a.ilim) {
        const int64_t hi = gen + 1 < a.maxiter ? gen + 1 : a.maxiter;
        for (int64_t k = gen - a.ilim + tid; k < hi; k += T) {
            const double v = a.besthist[k];
            wmax = fmax(wmax, v), wmin = fmin(wmin, v);
        }
    }
    for (int64_t k = tid; k < a.maxiter; k += T) {
        const double v = a.besthist[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    for (int64_t k = tid; k < a.P; k += T) {
        const double v = a.fit[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    {
        double v10[10] = {nv2n, any3, any6, fail8, nan_sd, sdmax, wmax, jmax, wmin, jmin};
        const int k10[10] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2};
        reduce_many<10, kVdWideWaves>(v10, k10, red10);
        nv2n = v10[0], any3 = v10[1], any6 = v10[2], fail8 = v10[3], nan_sd = v10[4], sdmax = v10[5], wmax = v10[6];
        jmax = v10[7], wmin = v10[8], jmin = v10[9];
    }
    const double nvn = sqrt(nv2n);
#pragma unroll 4
    for (int e = tid; e < n; e += T) a.vn[e] = a.vvec[e] / nvn;
    int status = SX_STATUS_NONE;
    if (gen >= a.maxiter)
        status = -1;
    else if (sqrt(dx2) <= a.xtol && fbest < a.ftol)
        status = 0;
    else if (fbest <= a.ftol)
        status = 1;
    else if (any3 > 0.0)
        status = -3;
    else if (gen >= a.ilim && wmax - wmin < 1.0e-10)
        status = -5;
    else if (any6 > 0.0)
        status = -6;
    else if (gen > 2 && jmax - jmin < 1.0e-12)
        status = -7;
    else if (fail8 == 0.0 && nan_sd == 0.0 && sigma * sdmax < 1.0e-11 * a.insigma)
        status = -8;
    if (status != SX_STATUS_NONE) {  // the caller's result: best candidate of THIS generation, un-standardised
        const double *row = a.arx + state->best_row * (int64_t)n;
        for (int e = tid; e < n; e += T) {
            double x = row[e];
            if (a.pen_ws != nullptr) x = fmin(fmax(x, -1.0), 1.0);
            a.xbest[e] = x * a.xstd[e] + a.xm[e];
        }
    }
    __syncthreads();
    if (tid == 0) {
        state->sigma = sigma;
        state->reserved[0] = ps;
        state->reserved[1] = nv2n;
        state->reserved[2] = nvn;
        state->reserved[3] = 1.0;
        state->reserved[4] = sqrt(1.0 + nv2n) - 1.0;
        state->it = gen;
        state->nfev = gen * a.P;
        if (status != SX_STATUS_NONE) {
            state->status = status;
            state->stop_it = gen;
            __threadfence();
            state->done = 1;
        }
    }
}

}  // namespace

namespace {
int check_vd_args(const sx_vd_args *a, int64_t gen) {
    SX_REQUIRE(a && a->Z && a->ary && a->arx && a->fit && a->xmean && a->xold && a->dx && a->dvec && a->vvec && a->vn &&
                   a->pc && a->zinj && a->dy && a->w && a->mws && a->mout && a->besthist && a->xm && a->xstd && a->xbest &&
                   a->order && a->state,
               "sx_vdcma_generation: null pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->mu >= 1 && a->mu <= a->P && gen >= 1 && gen <= a->maxiter,
               "sx_vdcma_generation: bad shape or generation number");
    return 0;
}

// candidates [row0, row0 + rows) of generation `gen` (the injected pair +-dy is rows 0 and 1 of the generation): normals
// keyed by the global row, the injection's own row and its dy (replicated), steps y, candidates x, objective (of the
// clipped points with Penalize) -> ary_out, arx_out (rows, n), fit_out (rows); a->Z: scratch (rows, n)
int vd_candidates(const sx_vd_args *a, int64_t gen, int64_t row0, int64_t rows, double *ary_out, double *arx_out,
                  double *fit_out, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    sx_cma_state *state = (sx_cma_state *)a->state;
    int rc;
    if ((rc = sx_cmaes_normals(a->Z, rows, n, row0, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    // the injection's own normal row: "row P" of the generation, one past the population (:245)
    if ((rc = sx_cmaes_normals(a->zinj, 1, n, a->P, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    if (n <= kVdPer * kVdThreads)
        hipLaunchKernelGGL(vd_inject_kernel, dim3(1), dim3(kVdThreads), 0, st, *a);
    else
        hipLaunchKernelGGL(vd_inject_wide_kernel, dim3(1), dim3(kVdWideThreads), 0, st, *a);
    if ((rc = sx::vd_sample_launch(a->Z, rows, n, a->dvec, a->vn, a->xmean, a->dy, ary_out, arx_out, state, stream, row0)))
        return rc;
    if (a->pen_ws == nullptr) return sx_eval(a->fun_id, arx_out, rows, n, n, a->xm, a->xstd, fit_out, nullptr, nullptr, stream);
    return sx_cmaes_eval_penalized(a->fun_id, arx_out, rows, n, a->xm, a->xstd, nullptr, fit_out, nullptr, stream);
}
int vd_model_update(const sx_vd_args *a, int64_t gen, void *stream);
}  // namespace

extern "C" int sx_vdcma_generation(const sx_vd_args *a, int64_t gen, void *stream) {
    if (int rc = check_vd_args(a, gen)) return rc;
    if (int rc = vd_candidates(a, gen, 0, a->P, a->ary, a->arx, a->fit, stream)) return rc;
    return vd_model_update(a, gen, stream);
}

// The same generation in two steps for candidates sharded over ranks (as sx_cmaes_generation_stage): stage 0 = this rank's
// candidates into ary_loc / arx_loc / fit_loc; the caller all-gathers them into a->ary / a->arx / a->fit; stage 1 = ranking,
// moments, the O(n) model update and the stop rules, replicated on every rank.
extern "C" int sx_vdcma_generation_stage(const sx_vd_args *a, int64_t gen, int stage, int64_t row0, int64_t rows,
                                         double *ary_loc, double *arx_loc, double *fit_loc, void *stream) {
    if (int rc = check_vd_args(a, gen)) return rc;
    if (stage == 0) {
        SX_REQUIRE(ary_loc && arx_loc && fit_loc && row0 >= 0 && rows >= 1 && row0 + rows <= a->P,
                   "sx_vdcma_generation_stage: bad shard");
        return vd_candidates(a, gen, row0, rows, ary_loc, arx_loc, fit_loc, stream);
    }
    return vd_model_update(a, gen, stream);
}

namespace {
int vd_model_update(const sx_vd_args *a, int64_t gen, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    const int64_t P = a->P;
    sx_cma_state *state = (sx_cma_state *)a->state;
    int rc;
    sx_cma_args h = {};  // what the kernels shared with CMA-ES read
    h.arx = a->arx, h.fit = a->fit, h.xm = a->xm, h.xstd = a->xstd, h.hist_x = a->hist_x, h.hist_f = a->hist_f;
    h.state = a->state, h.n = n, h.P = P, h.hist_rows = a->hist_rows, h.xmean = a->xmean, h.xold = a->xold;
    h.pen_ws = a->pen_ws, h.pen_order = a->pen_order, h.mueff = a->mueff, h.fun_id = a->fun_id;
    if (a->pen_ws != nullptr) {  // constraints="Penalize" (cmaes/_constraints.py:4-82, shared with CMA-ES): weights, excess
        if ((rc = sx::cma_penalize_launch(h, gen, a->dvec, a->vvec, stream))) return rc;
    }
    if ((rc = sx::cma_rank_launch(a->fit, P, a->order, state, a->besthist, gen, stream))) return rc;
    if (a->hist_x) {
        SX_REQUIRE(a->hist_f != nullptr && a->hist_rows >= 0 && a->hist_rows <= P, "sx_vdcma_generation: bad history arguments");
        if ((rc = sx::cma_history_launch(h, gen, stream))) return rc;
    }
    if ((rc = sx::vd_moments_launch(a->arx, a->ary, a->order, a->w, a->mu, n, a->dvec, a->vn, 0.0, state, a->mws, a->mout,
                                    stream)))
        return rc;
    if (n <= kVdPer * kVdThreads) {
        hipLaunchKernelGGL(vd_update_kernel, dim3(1), dim3(kVdThreads), 0, st, *a, gen);
    } else {  // scratch: the partial sums of the moments kernels (4 x 64 x n doubles behind t_k), free by now
        double *part = a->mws + ((a->mu + 7) / 8) * 8;
        hipLaunchKernelGGL(vd_update_wide_kernel, dim3(1), dim3(kVdWideThreads), 0, st, *a, gen, part, part + n);
    }
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace
