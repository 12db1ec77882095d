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
#include "sx_wide.hpp"

using namespace sx;

namespace sx {
int vd_sample_launch(const double *Z, int64_t P, int n, const double *dvec, const double *vn, const double *xmean,
                     const double *dy, double *ary, double *arx, const sx_cma_state *st, void *stream, int64_t row0 = 0);
int vd_moments_launch(const double *arx, const double *ary, const int64_t *idx, const double *w, int mu, int n,
                      const double *dvec, const double *vn, double norm_v2, const sx_cma_state *state, double *ws,
                      double *out, void *stream, const double *tk_rows = nullptr, const double *xmean = nullptr);
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
// Wide models (n > 4096, where VD-CMA matters most: vdcma/_vdcma.py:144-458 is the O(n) companion of CMA-ES).  The vectors
// no longer fit one workgroup's registers, and one workgroup walking them in memory costs 150 us at n = 16 384
// (profiles/r5_vd_wide.txt): the update is a CHAIN OF SMALL GRID-WIDE KERNELS instead, one per reduction of the reference's
// formulas.  Every kernel starts by adding up the previous one's per-workgroup partial sums (<= 256 of them, the same
// order in every workgroup: same bits everywhere), derives the scalars that follow from them, does its element-wise
// phase over the whole grid and leaves its own partials; p / q (later r, s, the natural-gradient steps) wait in two
// scratch vectors.  Same expressions, same association as vd_update_kernel; only the ORDER of the reductions' additions
// differs (both are compared with the oracle to rounding, tests/test_gpu_wide.py).
// The injection's normal row is drawn inside vd_inject_wide_kernel (only its norm is needed): no buffer, no launch.
// ---------------------------------------------------------------------------
constexpr int kVwThreads = 256;
constexpr int kVwWaves = kVwThreads / 64;
constexpr int kVwMaxBlocks = 256;
constexpr int kVwSlots = 10;  // partial sums per workgroup and phase
// scratch: [ sp (n) | sq (n) | part (8 phases x 256 workgroups x 10) | scal (64) ]
struct VwScratch {
    double *sp, *sq, *part, *scal;
};
__host__ __device__ inline VwScratch vw_scratch(double *base, int n) {
    VwScratch w;
    w.sp = base, w.sq = base + n, w.part = base + 2 * (int64_t)n;
    w.scal = w.part + 8 * kVwMaxBlocks * kVwSlots;
    return w;
}
// The eight barrier words of vw_chain_kernel: NOT in this scratch -- the moments kernels write their partial sums over it
// between the injection (which zeroes the words) and the chain -- but behind t_k in Z, the (P, n) buffer of which wide
// models use the first P doubles only.
constexpr int kVwSyncWords = 8;  // (seven barriers; a 64-byte line)
__host__ __device__ inline unsigned *vw_sync(const sx_vd_args &a) { return (unsigned *)(a.Z + a.P); }
inline unsigned vw_blocks(int n) {
    const int b = (n + kVwThreads - 1) / kVwThreads;
    return (unsigned)(b < kVwMaxBlocks ? b : kVwMaxBlocks);
}
// scal[]: 0 pos0, 1 pos1, 2 dx2, 3 vmax | 4 t, 5 svi | 6 vq | 7 ri | 8 svn | 9 g2, 10 mind | 16.. the state as the update
// found it: 16 sigma, 17 ps, 18 |v|^2, 19 |v|, 20 inject, 21 fbest, 22 best_row
enum { kSPos0 = 0, kSPos1, kSDx2, kSVmax, kST, kSSvi, kSVq, kSRi, kSSvn, kSG2, kSMind, kSState = 16 };

// What one workgroup leaves for the others inside vw_chain_kernel -- partial sums and scalars, nothing else: every vector
// element is written and read by the same thread in every phase (VW_FOR_E) -- goes through agent-scope accesses (sc1:
// written through, read past the XCD's L2) and the barrier only drains them: no fence.  (A release / acquire pair per
// barrier writes back and invalidates the XCD's whole L2 -- full of the candidates kernel's rows: the single launch was
// 10-25 us per generation SLOWER than nine launches that way, profiles/r6_vd_chain.txt.)
__device__ __forceinline__ double vw_ld(const double *p) {
    return __hip_atomic_load((__attribute__((address_space(1))) double *)const_cast<double *>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vw_st(double *p, double v) {
    __hip_atomic_store((__attribute__((address_space(1))) double *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the K partial sums of phase `ph` over all workgroups (kind: 0 sum, 1 max, 2 min), in every thread
template <int K>
__device__ __forceinline__ void vw_collect(const VwScratch &w, int ph, const int (&kind)[K], double (&v)[K], double (*red)[K]) {
    const double *p = w.part + ((int64_t)ph * kVwMaxBlocks + threadIdx.x) * kVwSlots;
#pragma unroll
    for (int q = 0; q < K; ++q)
        v[q] = threadIdx.x < gridDim.x ? vw_ld(p + q) : (kind[q] == 0 ? 0.0 : (kind[q] == 1 ? -__builtin_inf() : __builtin_inf()));
    reduce_many<K, kVwWaves>(v, kind, red);
}
template <int K>
__device__ __forceinline__ void vw_leave(const VwScratch &w, int ph, const int (&kind)[K], double (&v)[K], double (*red)[K]) {
    reduce_many<K, kVwWaves>(v, kind, red);
    if (threadIdx.x == 0) {
        double *p = w.part + ((int64_t)ph * kVwMaxBlocks + blockIdx.x) * kVwSlots;
#pragma unroll
        for (int q = 0; q < K; ++q) vw_st(p + q, v[q]);
    }
}

// scalars that follow from the state and the first reduction (:298-328)
struct VwModel {
    double sigma0, ps0, nv2, nv, fbest, ps, sigma, alpha, beta, bsca;
    bool inject, cond;
};
__device__ __forceinline__ VwModel vw_model(const sx_vd_args &a, const double *scal) {
    VwModel m;
    m.sigma0 = scal[kSState], m.ps0 = scal[kSState + 1], m.nv2 = scal[kSState + 2], m.nv = scal[kSState + 3];
    m.inject = scal[kSState + 4] != 0.0, m.fbest = scal[kSState + 5];
    const double vmax = scal[kSVmax];
    m.ps = m.ps0, m.sigma = m.sigma0, m.cond = true;
    if (m.inject) {
        const double gap = (scal[kSPos1] - scal[kSPos0]) / ((double)a.P - 1.0);
        m.ps = m.ps0 + a.cs * (gap - m.ps0);
        m.sigma = m.sigma0 * exp(m.ps / a.ds);
        m.cond = m.ps < 0.5;
    }
    const double gamma = 1.0 / sqrt(1.0 + m.nv2);
    m.alpha = sqrt(m.nv2 * m.nv2 + (1.0 + m.nv2) / vmax * (2.0 - gamma)) / (2.0 + m.nv2);
    m.beta = 0.0;
    if (m.alpha < 1.0) {
        const double t2 = 1.0 + 2.0 / m.nv2;
        m.beta = (4.0 - (2.0 - gamma) / vmax) / (t2 * t2);
    } else {
        m.alpha = 1.0;
    }
    m.bsca = 2.0 * (m.alpha * m.alpha) - m.beta;
    return m;
}
__device__ __forceinline__ double vw_avec(const VwModel &m, double vnn) { return 2.0 - (m.bsca + 2.0 * (m.alpha * m.alpha)) * vnn; }

#define VW_FOR_E for (int e = (int)(blockIdx.x * kVwThreads + threadIdx.x); e < n; e += (int)(gridDim.x * kVwThreads))

// the mean-shift injection (:241-247) with the injection's normals ("row P" of the generation) drawn in place
__global__ __launch_bounds__(kVwThreads) void vw_inject_norms_kernel(const sx_vd_args a, int64_t gen, double *base) {
    __shared__ double red3[kVwWaves][3];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (blockIdx.x == 0 && threadIdx.x < kVwSyncWords) vw_sync(a)[threadIdx.x] = 0u;  // (vw_chain_kernel's barrier words)
    if (state->done || state->reserved[3] == 0.0) return;
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    VW_FOR_E {
        const double ddx = a.dx[e] / a.dvec[e];
        s1 += ddx * ddx;
        s2 += ddx * a.vvec[e];
    }
    // |z|^2 of the normal row: pair j = 64 k + l covers the elements 128 k + l and 128 k + 64 + l (cma_normals_kernel)
    const int npair = ((n + 127) / 128) * 64;
    for (int j = (int)(blockIdx.x * kVwThreads + threadIdx.x); j < npair; j += (int)(gridDim.x * kVwThreads)) {
        const int e0 = (j >> 6) * 128 + (j & 63), e1 = e0 + 64;
        if (e0 >= n) continue;
        const U4 wd = philox4x32_10((uint32_t)j, (uint32_t)a.P, (uint32_t)gen, kPurposeCmaNormal, a.key0, a.key1);
        const double d0 = u53(wd.x, wd.y), d1 = u53(wd.z, wd.w);
        const double rad = sqrt(-2.0 * log(1.0 - d0));
        double sn, cs;
        sincos(6.283185307179586 * d1, &sn, &cs);
        const double z0 = rad * cs, z1 = rad * sn;
        s3 += z0 * z0;
        if (e1 < n) s3 += z1 * z1;
    }
    double v3[3] = {s1, s2, s3};
    const int k3[3] = {0, 0, 0};
    vw_leave<3>(w, 7, k3, v3, red3);
}
__global__ __launch_bounds__(kVwThreads) void vw_inject_apply_kernel(const sx_vd_args a, double *base) {
    __shared__ double red3[kVwWaves][3];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done || state->reserved[3] == 0.0) return;
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    double v3[3];
    const int k3[3] = {0, 0, 0};
    vw_collect<3>(w, 7, k3, v3, red3);
    const double mnorm = v3[0] - v3[1] * v3[1] / (1.0 + state->reserved[1]);
    const double fac = sqrt(v3[2]) / sqrt(mnorm);
    VW_FOR_E a.dy[e] = fac * a.dx[e];
}

// phase A: the injected pair in the ranking (:299-300), mean shift (:292-294)
__device__ __forceinline__ void vw_a_phase(const sx_vd_args &a, double *base) {
    __shared__ double red4[kVwWaves][4];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    const double *wx = a.mout;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the state as this update found it (the last kernel of the chain rewrites it)
        vw_st(w.scal + kSState, state->sigma), vw_st(w.scal + kSState + 1, state->reserved[0]);
        vw_st(w.scal + kSState + 2, state->reserved[1]), vw_st(w.scal + kSState + 3, state->reserved[2]);
        vw_st(w.scal + kSState + 4, state->reserved[3]), vw_st(w.scal + kSState + 5, state->fbest);
        vw_st(w.scal + kSState + 6, (double)state->best_row);
    }
    double pos0 = 0.0, pos1 = 0.0;
    if (state->reserved[3] != 0.0) {
        for (int64_t k = blockIdx.x * kVwThreads + threadIdx.x; k < a.P; k += (int64_t)gridDim.x * kVwThreads) {
            const int64_t r = a.order[k];
            if (r == 0) pos0 = (double)k;
            if (r == 1) pos1 = (double)k;
        }
    }
    double dx2 = 0.0, vmax = -__builtin_inf();
    VW_FOR_E {
        const double xm0 = a.xmean[e], v1 = a.vn[e];
        const double dxe = wx[e] - a.wsum * xm0;
        a.dx[e] = dxe;
        a.xold[e] = xm0;
        a.xmean[e] = xm0 + dxe;
        dx2 += dxe * dxe;
        vmax = fmax(vmax, v1 * v1);
    }
    double v4[4] = {pos0, pos1, dx2, vmax};
    const int k4[4] = {0, 0, 0, 1};
    vw_leave<4>(w, 0, k4, v4, red4);
}

// phase B: evolution path (:309-314), t = y . vn, sum of vn^2 invavnn
__device__ __forceinline__ void vw_b_phase(const sx_vd_args &a, double *base) {
    __shared__ double red4[kVwWaves][4];
    __shared__ double red2[kVwWaves][2];
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    double v4[4];
    const int k4[4] = {0, 0, 0, 1};
    vw_collect<4>(w, 0, k4, v4, red4);
    __shared__ double s_scal[32];
    __syncthreads();  // (one launch for the whole chain: the previous phase's readers of its s_scal are through)
    if (threadIdx.x < 4) s_scal[threadIdx.x] = v4[threadIdx.x];
    if (threadIdx.x >= kSState && threadIdx.x < kSState + 8) s_scal[threadIdx.x] = vw_ld(w.scal + threadIdx.x);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 4) vw_st(w.scal + threadIdx.x, v4[threadIdx.x]);
    const VwModel m = vw_model(a, s_scal);
    const double *wy = a.mout + n;
    const double cpc = sqrt(a.cc * (2.0 - a.cc) * a.mueff);
    double t = 0.0, svi = 0.0;
    VW_FOR_E {
        double pce = a.pc[e] * (1.0 - a.cc);
        if (m.cond) pce = pce + cpc * wy[e];
        a.pc[e] = pce;
        const double v1 = a.vn[e], y = pce / a.dvec[e];
        t += y * v1;
        const double vnn = v1 * v1;
        svi += vnn * (vnn / vw_avec(m, vnn));
    }
    double v2[2] = {t, svi};
    const int k2[2] = {0, 0};
    vw_leave<2>(w, 1, k2, v2, red2);
}

// phases C .. F (:331-378, :428-460): which = 0 p, q and vn . q;  1 r and r . invavnn;  2 s and s . vn^2;  3 the steps ngv, ngd
__device__ __forceinline__ void vw_cdef_phase(const sx_vd_args &a, double *base, const int which) {
    __shared__ double red2[kVwWaves][2];
    __shared__ double s_scal[32];
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    // the previous phase's sums (phase 1 + which: two values after B, one after C, D, E)
    double v2[2];
    const int k2[2] = {0, 0};
    vw_collect<2>(w, 1 + which, k2, v2, red2);
    __syncthreads();
    if (threadIdx.x < 32) s_scal[threadIdx.x] = vw_ld(w.scal + threadIdx.x);
    __syncthreads();
    const int slot = which == 0 ? kST : (which == 1 ? kSVq : (which == 2 ? kSRi : kSSvn));
    if (threadIdx.x == 0) {
        s_scal[slot] = v2[0];
        if (which == 0) s_scal[kSSvi] = v2[1];
        if (blockIdx.x == 0) {
            vw_st(w.scal + slot, v2[0]);
            if (which == 0) vw_st(w.scal + kSSvi, v2[1]);
        }
    }
    __syncthreads();
    const VwModel m = vw_model(a, s_scal);
    const double t = s_scal[kST], svi = s_scal[kSSvi], vq = s_scal[kSVq], ri = s_scal[kSRi], svn = s_scal[kSSvn];
    const double nv2 = m.nv2, nv = m.nv, alpha = m.alpha, bsca = m.bsca;
    const double *pmu = a.mout + 2 * (int64_t)n, *qmu = a.mout + 3 * (int64_t)n;
    const bool learn = a.cmu + a.c1 > 0.0;
    double acc = 0.0, mind = __builtin_inf();
    if (which == 0) {
        const double shrink = nv2 / (1.0 + nv2);
        VW_FOR_E {
            const double v1 = a.vn[e], y = a.pc[e] / a.dvec[e];
            double p = a.cmu != 0.0 ? a.cmu * pmu[e] : 0.0;
            double q = a.cmu != 0.0 ? a.cmu * qmu[e] : 0.0;
            if (m.cond && a.c1 != 0.0) {
                const double p_one = (y * y - shrink * ((t * y) * v1)) - 1.0;
                const double q_one = t * y - (0.5 * ((t * t + 1.0) + nv2)) * v1;
                p = p + a.c1 * p_one;
                q = q + a.c1 * q_one;
            }
            w.sp[e] = p, w.sq[e] = q;
            acc += v1 * q;
        }
    } else if (which == 1) {
        VW_FOR_E {
            const double v1 = a.vn[e], vnn = v1 * v1;
            const double r = w.sp[e] - alpha / (1.0 + nv2) * (((2.0 + nv2) * w.sq[e]) * v1 - (nv2 * vq) * vnn);
            w.sp[e] = r;
            acc += r * (vnn / vw_avec(m, vnn));
        }
    } else if (which == 2) {
        VW_FOR_E {
            const double v1 = a.vn[e], vnn = v1 * v1, av = vw_avec(m, vnn);
            const double sv = w.sp[e] / av - bsca * ri / (1.0 + bsca * svi) * (vnn / av);
            w.sp[e] = sv;
            acc += sv * vnn;
        }
    } else {
        VW_FOR_E {
            double qq = 0.0, pp = 0.0;
            if (learn) {
                const double v1 = a.vn[e], dv = a.dvec[e], sv = w.sp[e];
                qq = w.sq[e] / nv - alpha / nv * ((2.0 + nv2) * (v1 * sv) - svn * v1);
                pp = dv * sv;
                acc += qq * qq;
                mind = fmin(mind, dv / fabs(pp));
            }
            w.sq[e] = qq, w.sp[e] = pp;
        }
    }
    double o2[2] = {acc, mind};
    const int ko[2] = {0, 2};
    vw_leave<2>(w, 2 + which, ko, o2, red2);
}

// phase G: update of v and d (:371-378), the stopping rules' per-dimension counts, the best-fitness histories
__device__ __forceinline__ void vw_g_phase(const sx_vd_args &a, int64_t gen, double *base) {
    __shared__ double red2[kVwWaves][2];
    __shared__ double red10[kVwWaves][10];
    __shared__ double s_scal[32];
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    double v2[2];
    const int k2[2] = {0, 2};
    vw_collect<2>(w, 5, k2, v2, red2);
    __syncthreads();
    if (threadIdx.x < 32) s_scal[threadIdx.x] = vw_ld(w.scal + threadIdx.x);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) vw_st(w.scal + kSG2, v2[0]), vw_st(w.scal + kSMind, v2[1]);
    const VwModel m = vw_model(a, s_scal);
    const bool learn = a.cmu + a.c1 > 0.0;
    double up = 1.0;
    if (learn) {
        up = fmin(1.0, 0.7 * m.nv / sqrt(v2[0]));
        up = fmin(up, 0.7 * v2[1]);
    }
    const double sigma = m.sigma;
    double nv2n = 0.0, any3 = 0.0, any6 = 0.0, fail8 = 0.0, nan_sd = 0.0, sdmax = -__builtin_inf();
    VW_FOR_E {
        const double dv = a.dvec[e], vv0 = a.vvec[e];
        const double sd = sqrt((dv * (1.0 + vv0 * vv0)) * dv);  // sqrt of diag D (I + v v^T) D (:249-254)
        const double vv = vv0 + up * w.sq[e];
        a.vvec[e] = vv;
        a.dvec[e] = dv + up * w.sp[e];
        nv2n += vv * vv;
        if (0.2 * sigma * sd < 1.0e-10) any3 += 1.0;
        if (sigma * sd > 1.0e3 * a.insigma) any6 += 1.0;
        if (sd != sd) nan_sd += 1.0;
        sdmax = fmax(sdmax, sd);
        if (!(sigma * fabs(a.pc[e]) < 1.0e-11 * a.insigma)) fail8 += 1.0;
    }
    double wmax = -__builtin_inf(), wmin = __builtin_inf(), jmax = -__builtin_inf(), jmin = __builtin_inf();
    const int64_t g0 = blockIdx.x * kVwThreads + threadIdx.x, gs = (int64_t)gridDim.x * kVwThreads;
    if (gen >= a.ilim) {
        const int64_t hi = gen + 1 < a.maxiter ? gen + 1 : a.maxiter;
        for (int64_t k = gen - a.ilim + g0; k < hi; k += gs) {
            const double v = a.besthist[k];
            wmax = fmax(wmax, v), wmin = fmin(wmin, v);
        }
    }
    for (int64_t k = g0; k < a.maxiter; k += gs) {
        const double v = a.besthist[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    for (int64_t k = g0; k < a.P; k += gs) {
        const double v = a.fit[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    double v10[10] = {nv2n, any3, any6, fail8, nan_sd, sdmax, wmax, jmax, wmin, jmin};
    const int k10[10] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2};
    vw_leave<10>(w, 6, k10, v10, red10);
}

// phase H: vn = v / |v|, the stop rules (cmaes/_cmaes.py:360-434 as vdcma calls them), the result, the state
// (only workgroup 0 writes the state, at its very end, and nobody reads anything of it after phase A but `done` -- at the
//  launch's start; raise_done: one launch for the whole chain, where every workgroup has read `done` long before)
__device__ __forceinline__ void vw_h_phase(const sx_vd_args &a, int64_t gen, double *base, const bool raise_done) {
    __shared__ double red10[kVwWaves][10];
    __shared__ double s_scal[32];
    sx_cma_state *state = (sx_cma_state *)a.state;
    const int n = a.n;
    const VwScratch w = vw_scratch(base, n);
    double v10[10];
    const int k10[10] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2};
    vw_collect<10>(w, 6, k10, v10, red10);
    __syncthreads();
    if (threadIdx.x < 32) s_scal[threadIdx.x] = vw_ld(w.scal + threadIdx.x);
    __syncthreads();
    const VwModel m = vw_model(a, s_scal);
    const double nv2n = v10[0], any3 = v10[1], any6 = v10[2], fail8 = v10[3], nan_sd = v10[4], sdmax = v10[5], wmax = v10[6];
    const double jmax = v10[7], wmin = v10[8], jmin = v10[9];
    const double nvn = sqrt(nv2n), sigma = m.sigma, fbest = m.fbest, dx2 = s_scal[kSDx2];
    VW_FOR_E a.vn[e] = a.vvec[e] / nvn;
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
        // (arx NULL: the candidates were not kept; the row is formed again from its step -- old mean, old sigma -- as they were)
        const int64_t br = (int64_t)s_scal[kSState + 6] * (int64_t)n;
        VW_FOR_E {
            double x = a.arx != nullptr ? a.arx[br + e] : a.xold[e] + m.sigma0 * a.ary[br + e];
            if (a.pen_ws != nullptr) x = fmin(fmax(x, -1.0), 1.0);
            a.xbest[e] = x * a.xstd[e] + a.xm[e];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state->sigma = sigma;
        state->reserved[0] = m.ps;
        state->reserved[1] = nv2n;
        state->reserved[2] = nvn;
        state->reserved[3] = 1.0;  // from the second generation on the injection is on (:304-305)
        state->reserved[4] = sqrt(1.0 + nv2n) - 1.0;
        state->it = gen;
        state->nfev = gen * a.P;
        if (status != SX_STATUS_NONE) {
            state->status = status;
            state->stop_it = gen;
            if (raise_done) {
                __threadfence();
                state->done = 1;
            }
        }
    }
}
// One launch per phase (SX_VD_CHAIN=0, and what the single launch below is checked against) ...
__global__ __launch_bounds__(kVwThreads) void vw_a_kernel(const sx_vd_args a, double *base) {
    if (((const sx_cma_state *)a.state)->done) return;
    vw_a_phase(a, base);
}
__global__ __launch_bounds__(kVwThreads) void vw_b_kernel(const sx_vd_args a, double *base) {
    if (((const sx_cma_state *)a.state)->done) return;
    vw_b_phase(a, base);
}
__global__ __launch_bounds__(kVwThreads) void vw_cdef_kernel(const sx_vd_args a, double *base, const int which) {
    if (((const sx_cma_state *)a.state)->done) return;
    vw_cdef_phase(a, base, which);
}
__global__ __launch_bounds__(kVwThreads) void vw_g_kernel(const sx_vd_args a, int64_t gen, double *base) {
    if (((const sx_cma_state *)a.state)->done) return;
    vw_g_phase(a, gen, base);
}
__global__ __launch_bounds__(kVwThreads) void vw_h_kernel(const sx_vd_args a, int64_t gen, double *base) {
    if (((const sx_cma_state *)a.state)->done) return;
    vw_h_phase(a, gen, base, false);
}
// the `done` flag goes up in a launch of its own: every workgroup of vw_h_kernel has read the state by then
__global__ void vw_done_kernel(sx_cma_state *state) {
    if (state->status != SX_STATUS_NONE) state->done = 1;
}

// ... and ONE launch for the chain (round 6): the same phases on the same grid -- the same elements in the same workgroups,
// the same partial sums added in the same order: the same bits -- with a grid-wide barrier where a kernel boundary was.  A
// phase is ~1 us of work on <= 256 small workgroups, a kernel boundary ~5 us; a barrier -- every workgroup's first thread
// drains its write-through stores (vw_st), counts itself in at word k of the sync area and waits for the count to reach
// the grid's size -- is ~2 us on an idle chip.  The grid (<= 256 workgroups of 256 threads) is resident at once on any free device
// with >= 32 CUs; a wait that a foreign kernel stretches beyond kVwWaitTicks gives up: status kVwFault, done = 1 (the
// host raises).  The eight words are zeroed by vw_inject_norms_kernel, which every generation launches before its candidates.
constexpr long long kVwWaitTicks = 400000000LL;  // ~4 s of the 100 MHz wall clock
constexpr int kVwFault = -99;
__device__ __forceinline__ bool vw_grid_barrier(unsigned *word) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wavefront's partial sums / scalars have been written through)
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > kVwWaitTicks) {
                ok = 0;
                break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}
__global__ __launch_bounds__(kVwThreads) void vw_chain_kernel(const sx_vd_args a, int64_t gen, double *base, unsigned *sync) {
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    bool ok = true;
    vw_a_phase(a, base);
    ok = ok && vw_grid_barrier(sync + 0);
    if (ok) vw_b_phase(a, base);
    ok = ok && vw_grid_barrier(sync + 1);
#pragma unroll
    for (int which = 0; which < 4; ++which) {
        if (ok) vw_cdef_phase(a, base, which);
        ok = ok && vw_grid_barrier(sync + 2 + which);
    }
    if (ok) vw_g_phase(a, gen, base);
    ok = ok && vw_grid_barrier(sync + 6);
    if (ok) {
        vw_h_phase(a, gen, base, true);
    } else if (threadIdx.x == 0) {  // (whoever gave up says so; the others follow as their own waits run out or find the grid gone)
        state->status = kVwFault;
        state->stop_it = gen;
        __threadfence();
        state->done = 1;
    }
}

}  // namespace

namespace {
// wide models' scratch (two n-vectors, the phases' partial sums, the scalars): behind t_k in the moments workspace, whose
// 4 x 64 x n partial sums are dead whenever these kernels run (the injection before the candidates, the update after
// vd_moments_finish_kernel)
double *vd_wide_scratch(const sx_vd_args *a) { return a->mws + ((a->mu + 7) / 8) * 8; }

int check_vd_args(const sx_vd_args *a, int64_t gen) {
    SX_REQUIRE(a && a->Z && a->ary && a->fit && a->xmean && a->xold && a->dx && a->dvec && a->vvec && a->vn &&
                   a->pc && a->zinj && a->dy && a->w && a->mws && a->mout && a->besthist && a->xm && a->xstd && a->xbest &&
                   a->order && a->state,
               "sx_vdcma_generation: null pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->mu >= 1 && a->mu <= a->P && gen >= 1 && gen <= a->maxiter,
               "sx_vdcma_generation: bad shape or generation number");
    SX_REQUIRE(a->arx != nullptr || (a->n > kVdPer * kVdThreads && a->hist_x == nullptr && a->pen_ws == nullptr),
               "sx_vdcma_generation: arx may be NULL for wide models without history / Penalize only");
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
    if (n > kVdPer * kVdThreads) {
        // wide models: the injection's normals are drawn where their norm is needed, the candidates' normals where the
        // candidates are formed (sx_wide.hip wide_vd_candidates_kernel: normals, y, x, objective and t_k in one kernel);
        // t_k of all rows is kept (first P doubles of Z) when this call covers the whole generation
        double *ws = vd_wide_scratch(a);
        hipLaunchKernelGGL(vw_inject_norms_kernel, dim3(vw_blocks(n)), dim3(kVwThreads), 0, st, *a, gen, ws);
        hipLaunchKernelGGL(vw_inject_apply_kernel, dim3(vw_blocks(n)), dim3(kVwThreads), 0, st, *a, ws);
        SX_LAUNCH_CHECK();
        double *tk_rows = (row0 == 0 && rows == a->P && ary_out == a->ary) ? a->Z : nullptr;
        return wide_vd_candidates(a, gen, row0, rows, ary_out, arx_out, fit_out, tk_rows, st);
    }
    if ((rc = sx_cmaes_normals(a->Z, rows, n, row0, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    // the injection's own normal row: "row P" of the generation, one past the population (:245)
    if ((rc = sx_cmaes_normals(a->zinj, 1, n, a->P, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    hipLaunchKernelGGL(vd_inject_kernel, dim3(1), dim3(kVdThreads), 0, st, *a);
    if ((rc = sx::vd_sample_launch(a->Z, rows, n, a->dvec, a->vn, a->xmean, a->dy, ary_out, arx_out, state, stream, row0)))
        return rc;
    if (a->pen_ws == nullptr) return sx_eval(a->fun_id, arx_out, rows, n, n, a->xm, a->xstd, fit_out, nullptr, nullptr, stream);
    return sx_cmaes_eval_penalized(a->fun_id, arx_out, rows, n, a->xm, a->xstd, nullptr, fit_out, nullptr, stream);
}
int vd_model_update(const sx_vd_args *a, int64_t gen, void *stream, bool vd_wide_has_tk = false);
}  // namespace

extern "C" int sx_vdcma_generation(const sx_vd_args *a, int64_t gen, void *stream) {
    if (int rc = check_vd_args(a, gen)) return rc;
    if (int rc = vd_candidates(a, gen, 0, a->P, a->ary, a->arx, a->fit, stream)) return rc;
    return vd_model_update(a, gen, stream, true);
}

// The same generation in two steps for candidates sharded over ranks (as sx_cmaes_generation_stage): stage 0 = this rank's
// candidates into ary_loc / arx_loc / fit_loc; the caller all-gathers them into a->ary / a->arx / a->fit; stage 1 = ranking,
// moments, the O(n) model update and the stop rules, replicated on every rank.
extern "C" int sx_vdcma_generation_stage(const sx_vd_args *a, int64_t gen, int stage, int64_t row0, int64_t rows,
                                         double *ary_loc, double *arx_loc, double *fit_loc, void *stream) {
    if (int rc = check_vd_args(a, gen)) return rc;
    if (stage == 0) {
        SX_REQUIRE(ary_loc && (arx_loc || a->arx == nullptr) && fit_loc && row0 >= 0 && rows >= 1 && row0 + rows <= a->P,
                   "sx_vdcma_generation_stage: bad shard");
        return vd_candidates(a, gen, row0, rows, ary_loc, arx_loc, fit_loc, stream);
    }
    return vd_model_update(a, gen, stream);
}

namespace {
int vd_model_update(const sx_vd_args *a, int64_t gen, void *stream, bool vd_wide_has_tk) {
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
    const bool wide = n > kVdPer * kVdThreads;
    // (wide models, whole generation on this GPU: t_k of every row was left by the candidates kernel)
    const double *tk_rows = wide && vd_wide_has_tk ? a->Z : nullptr;
    if ((rc = sx::vd_moments_launch(a->arx, a->ary, a->order, a->w, a->mu, n, a->dvec, a->vn, 0.0, state, a->mws, a->mout,
                                    stream, tk_rows, a->xmean)))
        return rc;
    if (!wide) {
        hipLaunchKernelGGL(vd_update_kernel, dim3(1), dim3(kVdThreads), 0, st, *a, gen);
    } else {
        double *ws = vd_wide_scratch(a);
        const dim3 g(vw_blocks(n)), b(kVwThreads);
        static const bool one_launch = getenv("SX_VD_CHAIN") == nullptr || getenv("SX_VD_CHAIN")[0] != '0';
        if (one_launch) {
            hipLaunchKernelGGL(vw_chain_kernel, g, b, 0, st, *a, gen, ws, vw_sync(*a));
            SX_LAUNCH_CHECK();
            return 0;
        }
        hipLaunchKernelGGL(vw_a_kernel, g, b, 0, st, *a, ws);
        hipLaunchKernelGGL(vw_b_kernel, g, b, 0, st, *a, ws);
        for (int which = 0; which < 4; ++which) hipLaunchKernelGGL(vw_cdef_kernel, g, b, 0, st, *a, ws, which);
        hipLaunchKernelGGL(vw_g_kernel, g, b, 0, st, *a, gen, ws);
        hipLaunchKernelGGL(vw_h_kernel, g, b, 0, st, *a, gen, ws);
        hipLaunchKernelGGL(vw_done_kernel, dim3(1), dim3(1), 0, st, state);
    }
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace
