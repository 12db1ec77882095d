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
#include <algorithm>

#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace sx {
int cma_sample_launch(const double *xmean, double sigma, const double *sigma_p, const double *B, const double *D,
                      const double *Z, double *arx, int64_t P, int n, void *stream);
int cma_rank_mu_launch(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold, double sigma,
                       const double *sigma_p, const double *pc, double c1, double cmu, double tmp_coef,
                       const double *tmp_coef_p, double *C, double *ws_y, int n, void *stream, double *split_ws = nullptr,
                       int *mirrored = nullptr);
int eigh_enqueue(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes, int max_sweeps,
                 double tol, const int *skip, int refine, void *stream);  // sx_eigh.hip
int eigh_refine_in_loops();
int eigh_enqueue_phased(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes, double tol,
                        const int *skip, int refine, void *stream, int phases, int r0, int r1);  // sx_eigh.hip
int eigh_rounds_per_sweep(int n);
}  // namespace sx

namespace {

constexpr int kPartRows = 64;    // partial sums of the recombination
constexpr int kPathThreads = 1024;

static_assert(sizeof(sx_cma_state) == 128, "sx_cma_state is 128 bytes");

// numpy sorts NaN last: a < b in that order
__device__ __forceinline__ bool key_less(double a, double b) { return a < b || (b != b && a == a); }

// order = argsort(fit) (ties: lower index first); best row / value and the history entry of the generation.
// A wavefront ranks four elements: the keys pass through LDS 4096 at a time, lane l looks at keys l, l + 64, ... and the
// number of keys in front of an element is the population count of the wave's votes (one comparison per 64 keys and
// element instead of 64; round 3: 25.6 -> a few us at P = 1024, where the old form kept 16 workgroups busy with 256
// serial comparisons per thread).
// besthist == NULL: ranking only (the raw fitness behind Penalize's percentiles): no best row, no history entry
constexpr int kRankPerWave = 4;
__global__ __launch_bounds__(256) void cma_rank_kernel(const double *__restrict__ fit, int64_t P,
                                                       int64_t *__restrict__ order, sx_cma_state *state,
                                                       double *__restrict__ besthist, int64_t gen) {
    constexpr int CH = 4096;
    __shared__ double keys[CH];
    if (state->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * kRankPerWave;
    double fi[kRankPerWave];
    int cnt[kRankPerWave];
#pragma unroll
    for (int u = 0; u < kRankPerWave; ++u) fi[u] = i0 + u < P ? fit[i0 + u] : 0.0, cnt[u] = 0;
    for (int64_t c0 = 0; c0 < P; c0 += CH) {
        const int len = (int)(P - c0 < CH ? P - c0 : CH);
        __syncthreads();
        for (int e = threadIdx.x; e < len; e += 256) keys[e] = fit[c0 + e];
        __syncthreads();
        for (int k0 = 0; k0 < len; k0 += 64) {  // uniform trip count: every lane adds every vote
            const int k = k0 + lane;
            const bool in = k < len;
            const double fk = in ? keys[k] : 0.0;
            const int64_t kk = c0 + k;
#pragma unroll
            for (int u = 0; u < kRankPerWave; ++u) {
                const bool front = in && (key_less(fk, fi[u]) || (!key_less(fi[u], fk) && kk < i0 + u));
                cnt[u] += (int)__popcll(__ballot(front));
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kRankPerWave; ++u) {
        if (lane == u && i0 + u < P) {
            const int64_t rank = cnt[u];
            order[rank] = i0 + u;
            if (rank == 0 && besthist != nullptr) {
                state->best_row = i0 + u;
                state->fbest = fi[u];
                besthist[gen - 1] = fi[u];
            }
        }
    }
}
constexpr unsigned rank_grid(int64_t P) { return (unsigned)((P + 4 * kRankPerWave - 1) / (4 * kRankPerWave)); }

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

// return_all (cmaes/_cmaes.py:262-269): the first `rows` candidates of the generation, un-standardised, and their
// fitness -- or, with rows == 0, the generation's best candidate -- into the device-side history
__global__ __launch_bounds__(256) void cma_history_kernel(const sx_cma_args a, int64_t gen) {
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n;
    const int64_t rows = a.hist_rows > 0 ? a.hist_rows : 1;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * n) return;
    const int64_t r = t / n;
    const int e = (int)(t % n);
    const int64_t src = a.hist_rows > 0 ? r : state->best_row;
    double x = a.arx[src * n + e];
    if (a.pen_ws != nullptr) x = fmin(fmax(x, -1.0), 1.0);  // Penalize: the caller sees the clipped points (:238-256)
    a.hist_x[((gen - 1) * rows + r) * n + e] = x * a.xstd[e] + a.xm[e];
    if (e == 0) a.hist_f[(gen - 1) * rows + r] = a.fit[src];
}

// constraints="Penalize", the bookkeeping of cmaes/_constraints.py:33-76 on the device (round 3; the host-driven loop
// keeps its numpy form, optimize/_cmaes.py _BoundaryWeights): percentiles of the RAW fitness (np.percentile's linear rule
// incl. its lerp), the sliding history of fitness-spread estimates and its median, the weight growth for coordinates of
// the mean that sit outside the box, and v = weights / scale for the penalty pass.  One workgroup.
constexpr int kPenHist = 256;

__device__ __forceinline__ double np_lerp(double a, double b, double t) {  // numpy/lib/_function_base_impl.py _lerp
    const double d = b - a;
    return t >= 0.5 ? b - d * (1.0 - t) : a + d * t;
}
__device__ __forceinline__ double np_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : (x == 0.0 ? 0.0 : x)); }

// dvec / vvec != NULL: the model is VD-CMA's D (I + v v^T) D and its diagonal (d (1 + v^2)) d (vdcma/_vdcma.py:249-254)
__global__ __launch_bounds__(kPathThreads) void cma_penalty_kernel(const sx_cma_args a, int64_t gen,
                                                                   const double *__restrict__ dvec,
                                                                   const double *__restrict__ vvec) {
    __shared__ double red3[kPathThreads / 64][3];
    __shared__ double s_fill, s_meanlog;
    __shared__ int s_dofill, s_outside;
    __shared__ double s_sort[kPenHist];
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x;
    const int64_t P = a.P;
    double *w = a.pen_ws, *v = w + n, *hist = w + 2 * (int64_t)n + P, *meta = hist + kPenHist;
    const double sigma = state->sigma;
    // sums over the diagonal of C, their logarithms, and "any coordinate of the mean outside [-1, 1]"
    double sd = 0.0, sl = 0.0, so = 0.0;
    for (int e = tid; e < n; e += kPathThreads) {
        const double dc = dvec ? (dvec[e] * (1.0 + vvec[e] * vvec[e])) * dvec[e] : a.C[(int64_t)e * n + e], xm = a.xmean[e];
        sd += dc, sl += log(dc);
        if (xm < -1.0 || xm > 1.0) so += 1.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sd += __shfl_xor(sd, off, kWave), sl += __shfl_xor(sl, off, kWave), so += __shfl_xor(so, off, kWave);
    }
    if ((tid & 63) == 0) red3[tid >> 6][0] = sd, red3[tid >> 6][1] = sl, red3[tid >> 6][2] = so;
    __syncthreads();
    if (tid == 0) {
        sd = sl = so = 0.0;
        for (int k = 0; k < kPathThreads / 64; ++k) sd += red3[k][0], sl += red3[k][1], so += red3[k][2];
        // :34-35  q25, q75 = np.percentile(fit, [25, 75]): virtual index (P - 1) q, linear interpolation
        double q[2];
        for (int h = 0; h < 2; ++h) {
            const double vi = (double)(P - 1) * (h ? 0.75 : 0.25);
            const int64_t lo = (int64_t)floor(vi), hi = lo + 1 < P ? lo + 1 : P - 1;
            q[h] = np_lerp(a.fit[a.pen_order[lo]], a.fit[a.pen_order[hi]], vi - (double)lo);
        }
        double delta = (q[1] - q[0]) / (double)n / (sd / (double)n) / (sigma * sigma);
        int count = (int)meta[0], valid = (int)meta[1], ini = (int)meta[2];
        if (delta == 0.0) {  // :38-42
            double m = __builtin_huge_val();
            for (int k = 0; k < count; ++k)
                if (hist[k] > 0.0) m = fmin(m, hist[k]);
            delta = m;
        } else if (!valid) {
            count = 0, valid = 1;
        }
        const double cap = 20.0 + (3.0 * (double)n) / (double)P;  // :45-48 sliding window
        if ((double)count < cap) {
            hist[count++] = delta;
        } else {
            for (int k = 1; k < count; ++k) hist[k - 1] = hist[k];
            hist[count - 1] = delta;
        }
        const int outside = so > 0.0;
        int dofill = 0;
        if (ini && outside) {  // :56-59  weights = 2.0002 * median(history)
            for (int k = 0; k < count; ++k) {  // insertion sort of a copy
                const double x = hist[k];
                int j = k;
                while (j > 0 && s_sort[j - 1] > x) s_sort[j] = s_sort[j - 1], --j;
                s_sort[j] = x;
            }
            const double med = (count & 1) ? s_sort[count / 2] : 0.5 * (s_sort[count / 2 - 1] + s_sort[count / 2]);
            s_fill = 2.0002 * med;
            dofill = 1;
            if (valid && gen > 2) ini = 0;
        }
        meta[0] = (double)count, meta[1] = (double)valid, meta[2] = (double)ini;
        s_dofill = dofill, s_outside = outside, s_meanlog = sl / (double)n;
    }
    __syncthreads();
    const int dofill = s_dofill, outside = s_outside;
    const double fill = s_fill, meanlog = s_meanlog;
    const double kk = 3.0 * fmax(1.0, sqrt((double)n / a.mueff)) * sigma, fac = pow(1.2, fmin(1.0, a.mueff / 10.0 / (double)n));
    for (int e = tid; e < n; e += kPathThreads) {
        double we = dofill ? fill : w[e];
        const double dc = dvec ? (dvec[e] * (1.0 + vvec[e] * vvec[e])) * dvec[e] : a.C[(int64_t)e * n + e], xm = a.xmean[e];
        if (outside) {  // :61-73 (the excess is measured against a mean clipped on the UPPER side only, as the reference does)
            const double tx = xm - (xm > 1.0 ? 1.0 : xm);
            const bool out = xm < -1.0 || xm > 1.0;
            if (out && fabs(tx) > kk * sqrt(dc) && np_sign(tx) == np_sign(xm - a.xold[e])) we *= fac;
        }
        w[e] = we;
        v[e] = we / exp(0.9 * (log(dc) - meanlog));  // :76
    }
}

__global__ __launch_bounds__(256) void cma_add_penalty_kernel(const sx_cma_args a) {
    const sx_cma_state *state = (const sx_cma_state *)a.state;
    if (state->done) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.P) a.fit[i] = a.fit[i] + a.pen_ws[2 * (int64_t)a.n + i];  // :79
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

// The evolution-path step in three launches (the two products with B are spread over the chip):
//   cma_step_bt_kernel   xold = xmean; xmean = sum of the 64 partial rows (:273-274); step = xmean - xold;
//                        ypart[s][j] = sum over rows i of slice s of B[i][j] * step[i]          (grid n/64 x 8)
//   cma_b_y_kernel       y = (sum_s ypart[s]) / D;  isc = B y  = C^(-1/2) step                  (4 rows per workgroup)
//   cma_paths_kernel     ps, |ps|, cond, pc, sigma, tmp coefficient (:280-298), one workgroup
constexpr int kYSlices = 8;

__global__ __launch_bounds__(256) void cma_step_bt_kernel(const sx_cma_args a) {
    __shared__ double st[512];       // step of this workgroup's row slice (slices are <= 512 rows: n <= 4096)
    __shared__ double red[4][64];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int span = (n + kYSlices - 1) / kYSlices, r0 = blockIdx.y * span, r1 = r0 + span < n ? r0 + span : n;
    for (int i = r0 + tid; i < r1; i += 256) {
        const double xo = a.xmean[i];  // (every column block reads the same old mean: only block x = 0 replaces it, below)
        double xn = 0.0;
#pragma unroll 8
        for (int q = 0; q < kPartRows; ++q) xn += a.part[(int64_t)q * n + i];
        st[i - r0] = xn - xo;
        if (blockIdx.x == 0) a.step[i] = xn - xo, a.xold[i] = xo, a.xnew[i] = xn;
    }
    __syncthreads();
    const int col = blockIdx.x * 64 + tx;
    double acc = 0.0;
    if (col < n)
        for (int i = r0 + ty; i < r1; i += 4) acc += a.B[(int64_t)i * n + col] * st[i - r0];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && col < n) a.ypart[(int64_t)blockIdx.y * n + col] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

__global__ __launch_bounds__(256) void cma_b_y_kernel(const sx_cma_args a) {
    extern __shared__ double y[];  // n
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < n; j += 256) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < kYSlices; ++q) s += a.ypart[(int64_t)q * n + j];
        y[j] = s / a.D[j];
        if (blockIdx.x == 0) a.xmean[j] = a.xnew[j];  // the old mean has been consumed by every workgroup of the launch before
    }
    __syncthreads();
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += a.B[(int64_t)i * n + j] * y[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) a.isc[i] = acc;
}

__global__ __launch_bounds__(kPathThreads) void cma_paths_kernel(const sx_cma_args a, int64_t gen) {
    __shared__ double red[16];
    sx_cma_state *state = (sx_cma_state *)a.state;
    if (state->done) return;
    const int n = a.n, tid = threadIdx.x;
    const double sigma = state->sigma;
    // ps = (1-cs) ps + sqrt(cs (2-cs) mueff) * isc / sigma                                     :280-282
    const double kps = sqrt(a.cs * (2.0 - a.cs) * a.mueff);
    double q2 = 0.0;
    for (int e = tid; e < n; e += kPathThreads) {
        const double p = (1.0 - a.cs) * a.ps[e] + kps * a.isc[e] / sigma;
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
        if (cond) p += kpc * a.step[e] / sigma;                                                // :287
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
    __shared__ double red14[kPathThreads / 64][14];
    __shared__ double fin14[14];
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
    // one combined reduction: 7 sums, 4 maxima, 3 minima
    double vs[14] = {dx2, fail4, any5, any8, fail10, nan_sd, nan_d, dmax, sdmax, wmax, jmax, dmin, wmin, jmin};
#pragma unroll
    for (int q = 0; q < 14; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(vs[q], off, kWave);
            vs[q] = q < 7 ? vs[q] + o : (q < 11 ? fmax(vs[q], o) : fmin(vs[q], o));
        }
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 14; ++q) red14[tid >> 6][q] = vs[q];
    }
    __syncthreads();
    // the 16 waves' partials of quantity q are folded by 16 adjacent lanes (a workgroup-wide serial loop over
    // 14 x 16 LDS words in every thread was most of this kernel's 31 us at n = 512)
    static_assert(kPathThreads / 64 == 16, "sixteen partials per quantity");
    if (tid < 14 * 16) {
        const int q = tid >> 4;
        double r = red14[tid & 15][q];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double o = __shfl_xor(r, off, 16);
            r = q < 7 ? r + o : (q < 11 ? fmax(r, o) : fmin(r, o));
        }
        if ((tid & 15) == 0) fin14[q] = r;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 14; ++q) vs[q] = fin14[q];
    dx2 = vs[0], fail4 = vs[1], any5 = vs[2], any8 = vs[3], fail10 = vs[4], nan_sd = vs[5], nan_d = vs[6];
    dmax = vs[7], sdmax = vs[8], wmax = vs[9], jmax = vs[10], dmin = vs[11], wmin = vs[12], jmin = vs[13];
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
        for (int e = tid; e < n; e += kPathThreads) {
            double x = row[e];
            if (a.pen_ws != nullptr) x = fmin(fmax(x, -1.0), 1.0);  // Penalize: the clipped point (:336-350)
            a.xbest[e] = x * a.xstd[e] + a.xm[e];
        }
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

namespace sx {
// shared with the VD-CMA generation (sx_vd_loop.hip)
int cma_rank_launch(const double *fit, int64_t P, int64_t *order, sx_cma_state *state, double *besthist, int64_t gen,
                    void *stream) {
    hipLaunchKernelGGL(cma_rank_kernel, dim3(rank_grid(P)), dim3(256), 0, (hipStream_t)stream, fit, P, order,
                       state, besthist, gen);
    SX_LAUNCH_CHECK();
    return 0;
}
// Penalize for a loop that shares these kernels (sx_vd_loop.hip): h carries fit (raw, of the clipped candidates), arx, xm,
// xstd, xmean, xold, state, pen_ws, pen_order, n, P, mueff, fun_id; leaves the penalised fitness in h.fit
int cma_penalize_launch(const sx_cma_args &h, int64_t gen, const double *dvec, const double *vvec, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    SX_REQUIRE(h.pen_ws && h.pen_order && 20.0 + 3.0 * h.n / (double)h.P + 1.0 <= (double)kPenHist,
               "Penalize on the device needs pen_order and a spread history of at most 256 entries");
    hipLaunchKernelGGL(cma_rank_kernel, dim3(rank_grid(h.P)), dim3(256), 0, st, (const double *)h.fit, h.P, h.pen_order,
                       (sx_cma_state *)h.state, (double *)nullptr, gen);
    hipLaunchKernelGGL(cma_penalty_kernel, dim3(1), dim3(kPathThreads), 0, st, h, gen, dvec, vvec);
    if (int rc = sx_cmaes_eval_penalized(h.fun_id, h.arx, h.P, h.n, h.xm, h.xstd, h.pen_ws + h.n, h.fit,
                                         h.pen_ws + 2 * (int64_t)h.n, stream))
        return rc;
    hipLaunchKernelGGL(cma_add_penalty_kernel, dim3((unsigned)((h.P + 255) / 256)), dim3(256), 0, st, h);
    SX_LAUNCH_CHECK();
    return 0;
}
int cma_history_launch(const sx_cma_args &h, int64_t gen, void *stream) {
    const int64_t tot = (h.hist_rows > 0 ? h.hist_rows : 1) * h.n;
    hipLaunchKernelGGL(cma_history_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, gen);
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace sx

namespace {
int check_cma_args(const sx_cma_args *a, int64_t gen) {
    SX_REQUIRE(a && a->Z && a->arx && a->fit && a->xmean && a->xold && a->ps && a->pc && a->C && a->B && a->D && a->w &&
                   a->Y && a->part && a->step && a->isc && a->ypart && a->xnew && a->besthist && a->xm && a->xstd && a->xbest && a->eigw && a->order && a->state &&
                   a->eigh_ws,
               "sx_cmaes_generation: null pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->mu >= 1 && a->mu <= a->P && gen >= 1 && gen <= a->maxiter,
               "sx_cmaes_generation: bad shape or generation number");
    SX_REQUIRE(a->n <= 4096, "sx_cmaes_generation: n <= 4096");
    SX_REQUIRE(a->pen_ws == nullptr || (a->pen_order != nullptr && 20.0 + 3.0 * a->n / (double)a->P + 1.0 <= (double)kPenHist),
               "sx_cmaes_generation: Penalize needs pen_order and a spread history of at most 256 entries");
    return 0;
}

// candidates [row0, row0 + rows) of generation `gen`: normals (keyed by the GLOBAL row), sampling GEMM, objective (of the
// clipped points with Penalize) -> arx_out (rows, n), fit_out (rows); Z: scratch (rows, n)
int cma_candidates(const sx_cma_args *a, int64_t gen, int64_t row0, int64_t rows, double *Z, double *arx_out,
                   double *fit_out, void *stream) {
    const int n = a->n;
    sx_cma_state *state = (sx_cma_state *)a->state;
    int rc;
    if ((rc = sx_cmaes_normals(Z, rows, n, row0, (uint32_t)gen, a->key0, a->key1, stream))) return rc;
    if ((rc = sx::cma_sample_launch(a->xmean, 0.0, &state->sigma, a->B, a->D, Z, arx_out, rows, n, stream))) return rc;
    if (a->pen_ws == nullptr) return sx_eval(a->fun_id, arx_out, rows, n, n, a->xm, a->xstd, fit_out, nullptr, nullptr, stream);
    return sx_cmaes_eval_penalized(a->fun_id, arx_out, rows, n, a->xm, a->xstd, nullptr, fit_out, nullptr, stream);
}
int cma_model_update(const sx_cma_args *a, int64_t gen, int do_eigh, void *stream, int phase = -1, int r0 = 0, int r1 = 0);
}  // namespace

extern "C" int sx_cmaes_generation(const sx_cma_args *a, int64_t gen, int do_eigh, void *stream) {
    if (int rc = check_cma_args(a, gen)) return rc;
    if (int rc = cma_candidates(a, gen, 0, a->P, a->Z, a->arx, a->fit, stream)) return rc;
    return cma_model_update(a, gen, do_eigh, stream);
}

// The generation with its decomposition enqueued in pieces (one GPU; do_eigh != 0): phase 0 = candidates + model update up to
// and including the decomposition's start and its rounds [0, r1); the caller then reads the eigensolver's run record
// (sx_eigh_info / the head of eigh_ws) and either adds rounds (phase 1: [r0, r1)) or lets phase 2 finish the decomposition and
// apply the stop rules.  Same kernels in the same order as sx_cmaes_generation minus the launches behind the run's end.
// sx_eigh_rounds_per_sweep(n): rounds of one sweep (0: the one-workgroup solver of small n, which cannot be enqueued in pieces).
extern "C" int sx_eigh_rounds_per_sweep(int n) { return sx::eigh_rounds_per_sweep(n); }
extern "C" int sx_cmaes_generation_phased(const sx_cma_args *a, int64_t gen, int do_eigh, int phase, int r0, int r1,
                                          void *stream) {
    if (int rc = check_cma_args(a, gen)) return rc;
    SX_REQUIRE(do_eigh != 0 && phase >= 0 && phase <= 2 && r0 >= 0 && r1 >= r0 && sx::eigh_rounds_per_sweep(a->n) > 0,
               "sx_cmaes_generation_phased: bad arguments");
    if (phase == 0)
        if (int rc = cma_candidates(a, gen, 0, a->P, a->Z, a->arx, a->fit, stream)) return rc;
    return cma_model_update(a, gen, do_eigh, stream, phase, r0, r1);
}

// The same generation in two steps, for candidates sharded over ranks (workers > 1: what the reference's parallel backends
// shard, _common.py:58-72): stage 0 = this rank's candidates [row0, row0 + rows) into arx_loc / fit_loc (a->Z: the (P, n) buffer of the struct; stage 0 uses
// rows x n of it, the model update all of it as scratch); the caller all-gathers them into a->arx / a->fit; stage 1 = everything else, replicated on every rank
// (ranking, recombination, paths, covariance, decomposition, stop rules -- and Penalize's bookkeeping + penalty pass).
extern "C" int sx_cmaes_generation_stage(const sx_cma_args *a, int64_t gen, int do_eigh, int stage, int64_t row0,
                                         int64_t rows, double *arx_loc, double *fit_loc, void *stream) {
    if (int rc = check_cma_args(a, gen)) return rc;
    if (stage == 0) {
        SX_REQUIRE(arx_loc && fit_loc && row0 >= 0 && rows >= 1 && row0 + rows <= a->P, "sx_cmaes_generation_stage: bad shard");
        return cma_candidates(a, gen, row0, rows, a->Z, arx_loc, fit_loc, stream);
    }
    return cma_model_update(a, gen, do_eigh, stream);
}

namespace {
// phase -1: the whole model update.  A decomposition enqueued in pieces (sx_cmaes_generation_phased; the host looks at the
// eigensolver's run record between them instead of paying for a sweep of no-op launches "in case"): 0 = everything up to and
// including the rounds [0, r1) of the decomposition, 1 = the rounds [r0, r1), 2 = the decomposition's finish + the stop rules.
int cma_model_update(const sx_cma_args *a, int64_t gen, int do_eigh, void *stream, int phase, int r0, int r1) {
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    const int64_t P = a->P;
    sx_cma_state *state = (sx_cma_state *)a->state;
    int rc;
    // tolerance of the decomposition: what LAPACK's own guarantees, a backward error of n * eps * |C|_F (never below the
    // solver's default 1e-14): at n = 512 that is 5.7e-14 -- about one decomposition in two stops a sweep earlier
    const double tol = std::max(1.0e-14, (double)n * 1.1102230246251565e-16);
    if (phase >= 1) {
        SX_REQUIRE(do_eigh != 0, "sx_cmaes_generation_phased: phases 1 and 2 belong to a decomposition");
        if ((rc = sx::eigh_enqueue_phased(a->C, n, do_eigh == 2 ? a->B : nullptr, a->eigw, a->B, a->eigh_ws, a->eigh_ws_bytes, tol,
                                          &state->done, sx::eigh_refine_in_loops(), stream, phase == 1 ? 2 : 4, r0, r1)))
            return rc;
        if (phase == 2) {
            hipLaunchKernelGGL(cma_stop_kernel, dim3(1), dim3(kPathThreads), 0, st, *a, gen, 1);
            SX_LAUNCH_CHECK();
        }
        return 0;
    }
    if (a->pen_ws != nullptr) {
        // Penalize (cmaes/_constraints.py:4-82): a->fit holds the objective of the clipped candidates; the boundary-weight
        // bookkeeping from its percentiles, then the weighted squared excess on top (the second pass recomputes the same raw values)
        hipLaunchKernelGGL(cma_rank_kernel, dim3(rank_grid(P)), dim3(256), 0, st, a->fit, P, a->pen_order, state,
                           (double *)nullptr, gen);
        hipLaunchKernelGGL(cma_penalty_kernel, dim3(1), dim3(kPathThreads), 0, st, *a, gen, (const double *)nullptr,
                           (const double *)nullptr);
        if ((rc = sx_cmaes_eval_penalized(a->fun_id, a->arx, P, n, a->xm, a->xstd, a->pen_ws + n, a->fit,
                                          a->pen_ws + 2 * (int64_t)n, stream)))
            return rc;
        hipLaunchKernelGGL(cma_add_penalty_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, *a);
    }
    hipLaunchKernelGGL(cma_rank_kernel, dim3(rank_grid(P)), dim3(256), 0, st, a->fit, P, a->order, state,
                       a->besthist, gen);
    if (a->hist_x) {
        SX_REQUIRE(a->hist_f != nullptr && a->hist_rows >= 0 && a->hist_rows <= P, "sx_cmaes_generation: bad history arguments");
        const int64_t tot = (a->hist_rows > 0 ? a->hist_rows : 1) * n;
        hipLaunchKernelGGL(cma_history_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, *a, gen);
    }
    hipLaunchKernelGGL(cma_mean_partial_kernel, dim3((unsigned)((n + 63) / 64), kPartRows / 4), dim3(256), 0, st, a->arx,
                       a->order, a->w, a->mu, n, a->part);
    hipLaunchKernelGGL(cma_step_bt_kernel, dim3((unsigned)((n + 63) / 64), kYSlices), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(cma_b_y_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)n * sizeof(double), st, *a);
    hipLaunchKernelGGL(cma_paths_kernel, dim3(1), dim3(kPathThreads), 0, st, *a, gen);
    SX_LAUNCH_CHECK();
    // the normals (P x n) are dead by now: where they hold two n x n half-sums, the covariance update takes its
    // upper-triangle / split-K form (csrc/sx_cmaes.hip), which leaves C symmetric
    int mirrored = 0;
    double *split_ws = P >= 2 * (int64_t)n ? a->Z : nullptr;
    if ((rc = sx::cma_rank_mu_launch(a->arx, a->order, a->w, a->mu, a->xold, 0.0, &state->sigma, a->pc, a->c1, a->cmu, 0.0,
                                     &state->tmp_coef, a->C, a->Y, n, stream, split_ws, &mirrored)))
        return rc;
    if (do_eigh) {
        if (!mirrored && (rc = sx_symmetrize_upper(a->C, n, stream))) return rc;
        // do_eigh == 2: start from the previous eigenvectors (B is both the starting basis and the output)
        if (phase == 0) {
            return sx::eigh_enqueue_phased(a->C, n, do_eigh == 2 ? a->B : nullptr, a->eigw, a->B, a->eigh_ws, a->eigh_ws_bytes, tol,
                                           &state->done, sx::eigh_refine_in_loops(), stream, 1 | 2, 0, r1);
        }
        if ((rc = sx::eigh_enqueue(a->C, n, do_eigh == 2 ? a->B : nullptr, a->eigw, a->B, a->eigh_ws, a->eigh_ws_bytes,
                                   a->eig_sweeps, tol, &state->done, sx::eigh_refine_in_loops(), stream)))
            return rc;
    }
    hipLaunchKernelGGL(cma_stop_kernel, dim3(1), dim3(kPathThreads), 0, st, *a, gen, do_eigh ? 1 : 0);
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace
