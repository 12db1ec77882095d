// Neighbourhood Algorithm: the resampling step (the walk inside the Voronoi cells of the best models).
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/na/_na.py:265-305  mutation(): for every new sample i, start at model k = ix[i % nr] and,
//       axis by axis, draw uniformly between the cell walls along that axis:
//         d2  = ((U[:, 1:] - X[i, 1:]) ** 2).sum(axis=1)                              (:280, numpy's pairwise order)
//         lim = 0.5 * (Xall[k, j] + U[:, j] + (d1 - d2) / (Xall[k, j] - U[:, j]))      (:288)
//         low = max(lim[lim <= X[i, j]].max(), 0), high = min(lim[lim >= X[i, j]].min(), 1)   (:290-294)
//         X[i, j] = uniform(low, high)                                                (:296)
//         d2 += (U[:, j] - X[i, j]) ** 2 - (U[:, j + 1] - X[i, j + 1]) ** 2           (:301-303)
//       with U = all models but k.  Cost O(popsize * models * ndim) per generation: this is the part on the device.
// What stays on the host (optimize/_na.py): the ranking of all models (np.argsort, the reference's tie order), and the
// scalar d1 recurrence (:298-300) -- numpy SCALAR `** 2`, which is libm pow and not always the correctly rounded
// square; the walk is chaotic in those bits, so the host evaluates it with the same libm between two axis steps
// (one small D2H/H2D per axis and generation).
//
// Layout: the models are stored column-major, XT[l * cap + m] (axis l of model m; models appended generation by
// generation), so a sweep over the models reads contiguous memory; d2 is (popsize, cap) row-major.
#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace {

// numpy's pairwise add.reduce order for `count` terms (numpy/_core/src/umath/loops_utils.h.src):
// < 8 terms left to right from 0; up to 128 terms 8 running accumulators + tree + tail; above that split at
// (n/2) - (n/2) % 8, recursively.  In-thread, terms generated on the fly.
template <class F>
__device__ __forceinline__ double np_leaf_sum(const F &term, int lo, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int t = 0; t < n; ++t) r += term(lo + t);
        return r;
    }
    double r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = term(lo + u);
    int t = 8;
    for (; t < n - (n % 8); t += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] += term(lo + t + u);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; t < n; ++t) res += term(lo + t);
    return res;
}
template <class F>
__device__ double np_pairwise_sum(const F &term, int count) {
    if (count <= 128) return np_leaf_sum(term, 0, count);
    int lo[12], nn[12], stage[12];
    double left[12];
    int sp = 0;
    lo[0] = 0, nn[0] = count, stage[0] = 0;
    double val = 0.0;
    bool have = false;
    while (true) {
        if (!have) {
            if (nn[sp] <= 128) {
                val = np_leaf_sum(term, lo[sp], nn[sp]);
                have = true;
            } else {
                int n2 = nn[sp] / 2;
                n2 -= n2 % 8;
                stage[sp] = 1;
                lo[sp + 1] = lo[sp], nn[sp + 1] = n2, stage[sp + 1] = 0;
                ++sp;
            }
        } else {
            if (sp == 0) return val;
            --sp;  // deliver to the parent
            if (stage[sp] == 1) {
                left[sp] = val;
                stage[sp] = 2;
                int n2 = nn[sp] / 2;
                n2 -= n2 % 8;
                lo[sp + 1] = lo[sp] + n2, nn[sp + 1] = nn[sp] - n2, stage[sp + 1] = 0;
                ++sp;
                have = false;
            } else {
                val = left[sp] + val;
            }
        }
    }
}

// start of the walks of a generation: X[i] = model k_i; d2[i][m] = sum over axes 1.. of (XT[l][m] - X[i][l])^2
__global__ __launch_bounds__(256) void na_begin_kernel(const double *__restrict__ XT, int64_t cap, int64_t M, int n,
                                                       const int64_t *__restrict__ kidx, double *__restrict__ X,
                                                       double *__restrict__ d2) {
    extern __shared__ double centre[];  // n
    const int i = blockIdx.y;
    const int64_t k = kidx[i];
    for (int l = threadIdx.x; l < n; l += 256) {
        const double c = XT[(int64_t)l * cap + k];
        centre[l] = c;
        if (blockIdx.x == 0) X[(int64_t)i * n + l] = c;
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    auto term = [&](int t) {
        const double d = XT[(int64_t)(t + 1) * cap + m] - centre[t + 1];
        return d * d;
    };
    d2[(int64_t)i * cap + m] = np_pairwise_sum(term, n - 1);
}

// axis step j of every walk: the pending d2 update of the previous free axis jp (if any), then the cell walls
// along axis j: per workgroup the largest lim <= x and the smallest lim >= x over its slice of the models.
__global__ __launch_bounds__(256) void na_axis_kernel(const double *__restrict__ XT, int64_t cap, int64_t M, int n, int j,
                                                      int jp, const int64_t *__restrict__ kidx,
                                                      const double *__restrict__ X, const double *__restrict__ d1,
                                                      double *__restrict__ d2, double *__restrict__ part_lo,
                                                      double *__restrict__ part_hi) {
    __shared__ double slo[4], shi[4];
    const int i = blockIdx.y;
    const int64_t k = kidx[i];
    const double xj = X[(int64_t)i * n + j];  // still the centre's coordinate: Xall[k, j] == X[i, j]
    const double d1i = d1[i];
    double xp = 0.0, xp1 = 0.0;
    if (jp >= 0) xp = X[(int64_t)i * n + jp], xp1 = X[(int64_t)i * n + jp + 1];
    double lo = -__builtin_inf(), hi = __builtin_inf();
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        double dd = d2[(int64_t)i * cap + m];
        if (jp >= 0) {  // d2 += (U[:, jp] - X[i, jp]) ** 2 - (U[:, jp + 1] - X[i, jp + 1]) ** 2
            const double a = XT[(int64_t)jp * cap + m] - xp, b = XT[(int64_t)(jp + 1) * cap + m] - xp1;
            dd = dd + (a * a - b * b);
            d2[(int64_t)i * cap + m] = dd;
        }
        if (m == k) continue;  // U = np.delete(Xall, k)
        const double u = XT[(int64_t)j * cap + m];
        const double lim = 0.5 * ((xj + u) + (d1i - dd) / (xj - u));
        if (lim <= xj) lo = fmax(lo, lim);
        if (lim >= xj) hi = fmin(hi, lim);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmax(lo, __shfl_xor(lo, off, kWave));
        hi = fmin(hi, __shfl_xor(hi, off, kWave));
    }
    if ((threadIdx.x & 63) == 0) slo[threadIdx.x >> 6] = lo, shi[threadIdx.x >> 6] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        part_lo[(int64_t)i * gridDim.x + blockIdx.x] = fmax(fmax(slo[0], slo[1]), fmax(slo[2], slo[3]));
        part_hi[(int64_t)i * gridDim.x + blockIdx.x] = fmin(fmin(shi[0], shi[1]), fmin(shi[2], shi[3]));
    }
}

// X[i, j] = uniform(low, high) = low + (high - low) * u   (numpy's legacy uniform: loc + scale * double)
__global__ __launch_bounds__(64) void na_draw_kernel(const double *__restrict__ part_lo, const double *__restrict__ part_hi,
                                                     int nblk, int64_t P, int n, int j, const double *__restrict__ u,
                                                     double *__restrict__ X, double *__restrict__ xnew) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= P) return;
    double lo = -__builtin_inf(), hi = __builtin_inf();
    for (int b = 0; b < nblk; ++b) {
        lo = fmax(lo, part_lo[i * nblk + b]);
        hi = fmin(hi, part_hi[i * nblk + b]);
    }
    const double low = lo > 0.0 ? lo : 0.0;    // max(lim[idx].max(), 0.0), or 0.0 when no wall lies below
    const double high = hi < 1.0 ? hi : 1.0;   // min(lim[idx].min(), 1.0), or 1.0
    const double v = low + (high - low) * u[i * n + j];
    X[i * n + j] = v;
    xnew[i] = v;
}

// the finished samples: fixed axes -> 0 (:283-286), and the new columns of the model store
__global__ __launch_bounds__(256) void na_commit_kernel(double *__restrict__ X, int64_t P, int n,
                                                        const int32_t *__restrict__ fixed, double *__restrict__ XT,
                                                        int64_t cap, int64_t M) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= P * n) return;
    const int64_t i = t / n;
    const int l = (int)(t % n);
    double v = X[t];
    if (fixed[l]) v = 0.0, X[t] = 0.0;
    XT[(int64_t)l * cap + M + i] = v;
}

// Philox uniforms of a generation, the block layout of the row kernels (oracle/streams.py PhiloxStream.na_uniforms)
__global__ __launch_bounds__(256) void na_uniforms_kernel(double *__restrict__ U, int64_t P, int n, uint32_t gen,
                                                          uint32_t k0, uint32_t k1) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= P * n) return;
    const uint32_t row = (uint32_t)(t / n), e = (uint32_t)(t % n);
    const uint32_t lpr = (uint32_t)lanes_per_row(n);
    const uint32_t q = e / lpr, l = e & (lpr - 1u);
    const U4 w = philox4x32_10((q >> 1) * lpr + l, row, gen, kPurposeNaUniform, k0, k1);
    U[t] = (q & 1u) ? u53(w.z, w.w) : u53(w.x, w.y);
}

}  // namespace

extern "C" int sx_na_blocks(int64_t M) {
    const int64_t b = (M + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

extern "C" int sx_na_begin(const double *XT, int64_t cap, int64_t M, int n, const int64_t *kidx, int64_t P, double *X,
                           double *d2, void *stream) {
    SX_REQUIRE(XT && kidx && X && d2 && M >= 1 && M <= cap && n >= 1 && P >= 1, "sx_na_begin: bad arguments");
    hipLaunchKernelGGL(na_begin_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)P), dim3(256), (size_t)n * sizeof(double),
                       (hipStream_t)stream, XT, cap, M, n, kidx, X, d2);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_na_axis(const double *XT, int64_t cap, int64_t M, int n, int j, int jp, const int64_t *kidx, int64_t P,
                          const double *u, const double *d1, double *X, double *d2, double *ws, double *xnew, void *stream) {
    SX_REQUIRE(XT && kidx && u && d1 && X && d2 && ws && xnew && M >= 1 && M <= cap && P >= 1, "sx_na_axis: bad arguments");
    SX_REQUIRE(j >= 0 && j < n && jp >= -1 && jp < j && jp + 1 < n, "sx_na_axis: bad axis");
    const int nblk = sx_na_blocks(M);
    double *part_lo = ws, *part_hi = ws + (int64_t)P * nblk;
    hipLaunchKernelGGL(na_axis_kernel, dim3((unsigned)nblk, (unsigned)P), dim3(256), 0, (hipStream_t)stream, XT, cap, M, n, j,
                       jp, kidx, X, d1, d2, part_lo, part_hi);
    hipLaunchKernelGGL(na_draw_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, (hipStream_t)stream, part_lo, part_hi,
                       nblk, P, n, j, u, X, xnew);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_na_commit(double *X, int64_t P, int n, const int32_t *fixed, double *XT, int64_t cap, int64_t M,
                            void *stream) {
    SX_REQUIRE(X && fixed && XT && P >= 1 && n >= 1 && M >= 0 && M + P <= cap, "sx_na_commit: bad arguments");
    hipLaunchKernelGGL(na_commit_kernel, dim3((unsigned)((P * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, P, n,
                       fixed, XT, cap, M);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_na_uniforms(double *U, int64_t P, int n, uint32_t gen, uint32_t key0, uint32_t key1, void *stream) {
    SX_REQUIRE(U && P >= 1 && n >= 1, "sx_na_uniforms: bad arguments");
    hipLaunchKernelGGL(na_uniforms_kernel, dim3((unsigned)((P * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, U, P, n,
                       gen, key0, key1);
    SX_LAUNCH_CHECK();
    return 0;
}
