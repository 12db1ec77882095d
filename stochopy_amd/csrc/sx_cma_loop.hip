// CMA-ES, device-resident generation: everything between two looks of the host at the 128-byte state.
//
// Reference code replaced (paths relative to the reference checkout), on top of the kernels of sx_cmaes.hip
// (sampling :232-237, covariance update :290-295) and sx_eigh.hip (:303-305):
//   stochopy/optimize/cmaes/_cmaes.py:272-277  arindex = argsort(arfitness); xold = xmean; xmean = w @ arx[arindex[:mu]]
//   stochopy/optimize/cmaes/_cmaes.py:280-287  ps, cond, pc                  (evolution paths)
//   stochopy/optimize/cmaes/_cmaes.py:298      sigma *= exp((cs/damps)(|ps|/chind - 1))
//   stochopy/optimize/cmaes/_cmaes.py:306      D = sqrt(D)
//   stochopy/optimize/cmaes/_cmaes.py:360-434  converge: the ten ordered stopping rules, incl. the reads of the
//                                              zero-initialised history
// One generation = sx_cmaes_generation(): normals -> sample (MFMA) -> objective -> rank -> mean partials ->
// paths (one workgroup: mean, C^(-1/2) step as B((B^T step)/D), ps, cond, pc, sigma) -> covariance update (MFMA)
// -> [symmetrise + eigendecomposition] -> stop rules.  Step size, `cond` coefficient, best row, status live in
// sx_cma_state on the device; the host only decides WHEN the eigendecomposition is due (a function of the
// generation number) and looks at the state every few generations.  Once a stopping rule fires the result
// (best point of that generation, un-standardised) is copied aside and the bookkeeping kernels of later launches
// do nothing.
#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace sx {
int cma_sample_launch(const double *xmean, double sigma, const double *sigma_p, const double *B, const double *D,
                      const double *Z, double *arx, int64_t P, int n, void *stream);
int cma_rank_mu_launch(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold, double sigma,
                       const double *sigma_p, const double *pc, double c1, double cmu, double tmp_coef,
                       const double *tmp_coef_p, double *C, double *ws_y, int n, void *stream);
}  // namespace sx

namespace {

constexpr int kPartRows = 64;    // partial sums of the recombination
constexpr int kPathThreads = 1024;

static_assert(sizeof(sx_cma_state) == 128, "sx_cma_state is 128 bytes");

// numpy sorts NaN last: a < b in that order
__device__ __forceinline__ bool key_less(double a, double b) { return a < b || (b != b && a == a); }

// order = argsort(fit) (ties: lower index first); best row / value and the history entry of the generation
__global__ __launch_bounds__(256) void cma_rank_kernel(const double *__restrict__ fit, int64_t P,
                                                       int64_t *__restrict__ order, sx_cma_state *state,
                                                       double *__restrict__ besthist, int64_t gen) {
    __shared__ int part[4][64];
    if (state->done) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + tx;
    const double fi = i < P ? fit[i] : 0.0;
    int cnt = 0;
    if (i < P) {
        const int64_t span = (P + 3) / 4, k0 = ty * span, k1 = k0 + span < P ? k0 + span : P;
        for (int64_t k = k0; k < k1; ++k) {
            const double fk = fit[k];
            cnt += (key_less(fk, fi) || (!key_less(fi, fk) && k < i)) ? 1 : 0;
        }
    }
    part[ty][tx] = cnt;
    __syncthreads();
    if (ty == 0 && i < P) {
        const int64_t rank = (int64_t)part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
        order[rank] = i;
        if (rank == 0) {
            state->best_row = i;
            state->fbest = fi;
            besthist[gen - 1] = fi;
        }
    }
}

// part[q][e] = sum over k = q (mod 64) of w[k] * arx[order[k]][e]   (grid: ceil(n/64) x 16, 256 threads)
__global__ __launch_bounds__(256) void cma_mean_partial_kernel(const double *__restrict__ arx,
                                                               const int64_t *__restrict__ order,
                                                               const double *__restrict__ w, int mu, int n,
                                                               double *__restrict__ part) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    const int q = blockIdx.y * 4 + ty;
    if (col >= n) return;
    double acc = 0.0;
    for (int k0 = q; k0 < mu; k0 += 4 * kPartRows) {
        double v[4], ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * kPartRows;
            const bool in = k < mu;
            ww[u] = in ? w[k] : 0.0;
            v[u] = in ? arx[order[k] * (int64_t)n + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += ww[u] * v[u];
    }
    part[(int64_t)q * n + col] = acc;
}

template <class F>
__device__ double block_reduce(double v, double *red, F op) {  // all threads get the result; 1024 threads
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, kWave));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < kPathThreads / 64; ++k) s = op(s, red[k]);
    return s;
}

// mean, step, C^(-1/2) step = B ((B^T step) / D), evolution paths, cond, step size.  One workgroup.
__global__ __launch_bounds__(kPathThreads) void cma_paths_kernel(const sx_cma_args a, int64_t gen) {
    extern __shared__ double lds[];  // step[n] | y[n] | isc[n] | slices[16][64]
    __shared__ double red[16];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *step = lds, *y = lds + n, *isc = lds + 2 * n, *sl = lds + 3 * n;
    const double sigma = state->sigma;
    // xold = xmean; xmean = w @ arx[order[:mu]] (sum of the 64 partial rows, fixed order)   :273-274
    for (int e = tid; e < n; e += kPathThreads) {
        const double xo = a.xmean[e];
        double xn = 0.0;
#pragma unroll 8
        for (int q = 0; q < kPartRows; ++q) xn += a.part[(int64_t)q * n + e];
        a.xold[e] = xo;
        a.xmean[e] = xn;
        step[e] = xn - xo;
    }
    __syncthreads();
    // y = (B^T step) / D: 64 columns at a time, 16 row slices
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int col = c0 + lane;
        double acc = 0.0;
        if (col < n)
            for (int i = wave; i < n; i += 16) acc += a.B[(int64_t)i * n + col] * step[i];
        sl[wave * 64 + lane] = acc;
        __syncthreads();
        if (wave == 0 && col < n) {
            double s = sl[lane];
#pragma unroll
            for (int k = 1; k < 16; ++k) s += sl[k * 64 + lane];
            y[col] = s / a.D[col];
        }
        __syncthreads();
    }
    // isc = B y: one wavefront per row
    for (int i = wave; i < n; i += 16) {
        double acc = 0.0;
        for (int j = lane; j < n; j += 64) acc += a.B[(int64_t)i * n + j] * y[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
        if (lane == 0) isc[i] = acc;
    }
    __syncthreads();
    // ps = (1-cs) ps + sqrt(cs (2-cs) mueff) * isc / sigma                                     :280-282
    const double kps = sqrt(a.cs * (2.0 - a.cs) * a.mueff);
    double q2 = 0.0;
    for (int e = tid; e < n; e += kPathThreads) {
        const double p = (1.0 - a.cs) * a.ps[e] + kps * isc[e] / sigma;
        a.ps[e] = p;
        q2 += p * p;
    }
    q2 = block_reduce(q2, red, [](double u, double v) { return u + v; });
    const double psn = sqrt(q2);
    // cond = |ps| / sqrt(1 - (1-cs)^(2 nfev / P)) / chind < 1.4 + 2/(n+1)   with nfev = gen * P      :283-285
    const bool cond = psn / sqrt(1.0 - pow(1.0 - a.cs, 2.0 * (double)gen)) / a.chind < 1.4 + 2.0 / (n + 1.0);
    const double kpc = sqrt(a.cc * (2.0 - a.cc) * a.mueff);
    for (int e = tid; e < n; e += kPathThreads) {
        double p = a.pc[e] * (1.0 - a.cc);                                                     // :286
        if (cond) p += kpc * step[e] / sigma;                                                  // :287
        a.pc[e] = p;
    }
    if (tid == 0) {
        state->tmp_coef = cond ? 0.0 : a.c1 * a.cc * (2.0 - a.cc);                             // :291
        state->sigma_next = sigma * exp((a.cs / a.damps) * (psn / a.chind - 1.0));            // :298
        state->psnorm = psn;
    }
}

// D = sqrt(eigenvalues) when a decomposition was made, then the ten ordered stopping rules (:360-434), the
// result copy when one fires, and the publication of the new step size / generation counter.  One workgroup.
__global__ __launch_bounds__(kPathThreads) void cma_stop_kernel(const sx_cma_args a, int64_t gen, int did_eigh) {
    __shared__ double red[16];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x;
    if (did_eigh) {
        for (int e = tid; e < n; e += kPathThreads) a.D[e] = sqrt(a.eigw[e]);                  // :306
        __syncthreads();
    }
    const double sigma = state->sigma_next, fbest = state->fbest;
    const int axis = (int)(gen % n);
    const double dax = a.D[axis];
    auto fsum = [](double u, double v) { return u + v; };
    auto fmx = [](double u, double v) { return fmax(u, v); };
    auto fmn = [](double u, double v) { return fmin(u, v); };
    // per-dimension quantities.  "all(x < t)" is carried as the count of elements that FAIL (NaN fails, as in numpy)
    double dx2 = 0.0, fail4 = 0.0, any5 = 0.0, dmax = -__builtin_inf(), dmin = __builtin_inf(), any8 = 0.0;
    double sdmax = -__builtin_inf(), fail10 = 0.0, nan_sd = 0.0, nan_d = 0.0;
    for (int e = tid; e < n; e += kPathThreads) {
        const double d = a.xold[e] - a.xmean[e];
        dx2 += d * d;
        if (!(fabs(0.1 * sigma * a.B[(int64_t)e * n + axis] * dax) < 1.0e-10)) fail4 += 1.0;
        const double sd = sqrt(a.C[(int64_t)e * n + e]);
        if (0.2 * sigma * sd < 1.0e-10) any5 += 1.0;
        const double de = a.D[e];
        dmax = fmax(dmax, de), dmin = fmin(dmin, de);  // (numpy's max/min propagate NaN; rule 6 then compares False either way)
        if (de != de) nan_d += 1.0;
        if (sigma * sd > 1.0e3 * a.insigma) any8 += 1.0;
        if (sd != sd) nan_sd += 1.0;
        sdmax = fmax(sdmax, sd);
        if (!(sigma * fabs(a.pc[e]) < 1.0e-11 * a.insigma)) fail10 += 1.0;
    }
    dx2 = block_reduce(dx2, red, fsum);
    fail4 = block_reduce(fail4, red, fsum);
    any5 = block_reduce(any5, red, fsum);
    dmax = block_reduce(dmax, red, fmx);
    dmin = block_reduce(dmin, red, fmn);
    any8 = block_reduce(any8, red, fsum);
    sdmax = block_reduce(sdmax, red, fmx);
    fail10 = block_reduce(fail10, red, fsum);
    nan_sd = block_reduce(nan_sd, red, fsum);
    nan_d = block_reduce(nan_d, red, fsum);
    // histories: window [gen-ilim, gen] of the zero-initialised best-fitness array (entry `gen` is not written yet),
    // and the whole array joined with this generation's fitness values
    double wmax = -__builtin_inf(), wmin = __builtin_inf(), jmax = -__builtin_inf(), jmin = __builtin_inf();
    if (gen >= a.ilim) {
        const int64_t hi = gen + 1 < a.maxiter ? gen + 1 : a.maxiter;
        for (int64_t k = gen - a.ilim + tid; k < hi; k += kPathThreads) {
            const double v = a.besthist[k];
            wmax = fmax(wmax, v), wmin = fmin(wmin, v);
        }
    }
    for (int64_t k = tid; k < a.maxiter; k += kPathThreads) {
        const double v = a.besthist[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    for (int64_t k = tid; k < a.P; k += kPathThreads) {
        const double v = a.fit[k];
        jmax = fmax(jmax, v), jmin = fmin(jmin, v);
    }
    wmax = block_reduce(wmax, red, fmx);
    wmin = block_reduce(wmin, red, fmn);
    jmax = block_reduce(jmax, red, fmx);
    jmin = block_reduce(jmin, red, fmn);
    int status = SX_STATUS_NONE;
    if (gen >= a.maxiter)
        status = -1;
    else if (sqrt(dx2) <= a.xtol && fbest < a.ftol)
        status = 0;
    else if (fbest <= a.ftol)
        status = 1;
    else if (fail4 == 0.0)
        status = -2;
    else if (any5 > 0.0)
        status = -3;
    else if (nan_d == 0.0 && dmax > 1.0e7 * dmin)
        status = -4;
    else if (gen >= a.ilim && wmax - wmin < 1.0e-10)
        status = -5;
    else if (any8 > 0.0)
        status = -6;
    else if (gen > 2 && jmax - jmin < 1.0e-12)
        status = -7;
    else if (fail10 == 0.0 && nan_sd == 0.0 && sigma * sdmax < 1.0e-11 * a.insigma)
        status = -8;
    if (status != SX_STATUS_NONE) {  // the caller's result: best candidate of THIS generation, un-standardised (:345-353)
        const double *row = a.arx + state->best_row * (int64_t)n;
        for (int e = tid; e < n; e += kPathThreads) a.xbest[e] = row[e] * a.xstd[e] + a.xm[e];
    }
    __syncthreads();
    if (tid == 0) {
        state->sigma = sigma;
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

extern "C" int sx_cmaes_generation(const sx_cma_args *a, int64_t gen, int do_eigh, void *stream) {
    SX_REQUIRE(a && a->Z && a->arx && a->fit && a->xmean && a->xold && a->ps && a->pc && a->C && a->B && a->D && a->w &&
                   a->Y && a->part && a->besthist && a->xm && a->xstd && a->xbest && a->eigw && a->order && a->state &&
                   a->eigh_ws,
               "sx_cmaes_generation: null pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->mu >= 1 && a->mu <= a->P && gen >= 1 && gen <= a->maxiter,
               "sx_cmaes_generation: bad shape or generation number");
    SX_REQUIRE(a->n <= 4096, "sx_cmaes_generation: n <= 4096");
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    const int64_t P = a->P;
    sx_cma_state *state = (sx_cma_state *)a->state;
    int rc;
    if ((rc = sx_cmaes_normals(a->Z, P, n, 0, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    if ((rc = sx::cma_sample_launch(a->xmean, 0.0, &state->sigma, a->B, a->D, a->Z, a->arx, P, n, stream))) return rc;
    if ((rc = sx_eval(a->fun_id, a->arx, P, n, n, a->xm, a->xstd, a->fit, nullptr, nullptr, stream))) return rc;
    hipLaunchKernelGGL(cma_rank_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, st, a->fit, P, a->order, state,
                       a->besthist, gen);
    hipLaunchKernelGGL(cma_mean_partial_kernel, dim3((unsigned)((n + 63) / 64), kPartRows / 4), dim3(256), 0, st, a->arx,
                       a->order, a->w, a->mu, n, a->part);
    const size_t lds = (size_t)(3 * n + 16 * 64) * sizeof(double);
    if (lds > 48 * 1024) {
        static bool raised = false;  // rows beyond ~1700 elements need more than the default dynamic-LDS limit
        if (!raised) {
            SX_HIP(hipFuncSetAttribute((const void *)cma_paths_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            raised = true;
        }
    }
    hipLaunchKernelGGL(cma_paths_kernel, dim3(1), dim3(kPathThreads), lds, st, *a, gen);
    SX_LAUNCH_CHECK();
    if ((rc = sx::cma_rank_mu_launch(a->arx, a->order, a->w, a->mu, a->xold, 0.0, &state->sigma, a->pc, a->c1, a->cmu, 0.0,
                                     &state->tmp_coef, a->C, a->Y, n, stream)))
        return rc;
    if (do_eigh) {
        if ((rc = sx_symmetrize_upper(a->C, n, stream))) return rc;
        // do_eigh == 2: start from the previous eigenvectors (B is both the starting basis and the output)
        if ((rc = sx_eigh(a->C, n, do_eigh == 2 ? a->B : nullptr, a->eigw, a->B, a->eigh_ws, a->eigh_ws_bytes,
                          a->eig_sweeps, 0.0, stream)))
            return rc;
    }
    hipLaunchKernelGGL(cma_stop_kernel, dim3(1), dim3(kPathThreads), 0, st, *a, gen, do_eigh ? 1 : 0);
    SX_LAUNCH_CHECK();
    return 0;
}
