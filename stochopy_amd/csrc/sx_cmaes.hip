// CMA-ES device kernels: sampling (m + sigma * B * (D o z)) and the covariance
// update (rank-mu + rank-one) as LDS-tiled fp64 MFMA contractions, the weighted
// recombination of the mean, Philox normals, and two small helpers.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/cmaes/_cmaes.py:232-237  arx[i] = xmean + sigma * dot(B, D * randn(n))
//   stochopy/optimize/cmaes/_cmaes.py:274      xmean = dot(weights, arx[arindex[:mu]])
//   stochopy/optimize/cmaes/_cmaes.py:290-295  artmp, C *= 1-c1-cmu; C += cmu*A^T diag(w) A; C += c1*pc pc^T; C += tmp
//   stochopy/optimize/cmaes/_cmaes.py:303      C = triu(C) + triu(C,1).T
//
// MFMA: v_mfma_f64_16x16x4_f64.  Operand layout (cdna_hip_programming.md section 3):
//   A (16x4): lane l holds A[l & 15][l >> 4];  B (4x16): lane l holds B[l >> 4][l & 15];
//   C/D: 4 doubles per lane, col = l & 15, row = (l >> 4) + 4 * reg.
// Workgroup tile 64x64, four waves in a 2x2 grid of 32x32 (2x2 MFMA tiles each), K chunk 32.
#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, KC = 32;
constexpr int LDA = KC + 2;   // A tile [BM][LDA]: row stride 34 doubles -> conflict-free ds_read_b64 of a column slab
constexpr int LDB = BN + 16;  // B tile [KC][LDB]: the two k-groups of a 32-lane half land 32 banks apart
constexpr int kGemmThreads = 256;

struct SampleOp {  // arx = xmean + sigma * (Z o D) * B^T
    const double *Z;      // (P,n)
    const double *Bm;     // (n,n) eigenvectors, row-major
    const double *D;      // (n)
    const double *xmean;  // (n)
    double *arx;          // (P,n)
    double sigma;
    int64_t P;
    int n;
};

struct RankMuOp {  // C = (1-c1-cmu)*C + cmu * Y^T diag(w) Y + c1 * pc pc^T + tmpc * C_old,  Y[k] = (arx[idx[k]] - xold)/sigma
    const double *arx;    // (P,n)
    const int64_t *idx;   // (mu) selected rows, best first
    const double *w;      // (mu)
    const double *xold;   // (n)
    const double *pc;     // (n)
    double *C;            // (n,n) in place
    double sigma, decay, cmu, c1, tmpc;
    int mu, n;
};

// MODE 0: M = P rows, N = n, K = n.   MODE 1: M = N = n, K = mu.
template <int MODE, class Op>
__global__ __launch_bounds__(kGemmThreads) void cma_gemm_kernel(const Op op) {
    __shared__ __attribute__((aligned(16))) double As[BM * LDA];
    __shared__ __attribute__((aligned(16))) double Bs[KC * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;  // wave tile origin inside the workgroup tile
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    int64_t M;
    int N, K;
    if (MODE == 0) {
        const SampleOp &o = (const SampleOp &)op;
        M = o.P;
        N = o.n;
        K = o.n;
    } else {
        const RankMuOp &o = (const RankMuOp &)op;
        M = o.n;
        N = o.n;
        K = o.mu;
    }
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < K; k0 += KC) {
        // ---- stage the A tile As[i][k] and the B tile Bs[k][j] ----
        if (MODE == 0) {
            const SampleOp &o = (const SampleOp &)op;
            // A[i][k] = Z[m0+i][k0+k] * D[k0+k]: thread -> row tid/4, 8 contiguous k
            {
                const int i = tid >> 2, kk = (tid & 3) * 8;
                const int64_t gi = m0 + i;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int gk = k0 + kk + u;
                    double v = 0.0;
                    if (gi < M && gk < K) v = o.D[gk] * o.Z[gi * (int64_t)o.n + gk];  // D * z  (:234)
                    As[i * LDA + kk + u] = v;
                }
            }
            // B[k][j] = Bm[n0+j][k0+k]: thread -> row j = tid/4, 8 contiguous k, transposed store
            {
                const int j = tid >> 2, kk = (tid & 3) * 8;
                const int gj = n0 + j;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int gk = k0 + kk + u;
                    double v = 0.0;
                    if (gj < N && gk < K) v = o.Bm[(int64_t)gj * o.n + gk];
                    Bs[(kk + u) * LDB + j] = v;
                }
            }
        } else {
            const RankMuOp &o = (const RankMuOp &)op;
            // selected row k: y = (arx[idx[k]] - xold) / sigma (:290); A[i][k] = y[m0+i] * w[k]; B[k][j] = y[n0+j]
            // thread -> k = tid/8 (0..31), 8 contiguous columns starting at (tid&7)*8
            const int kk = tid >> 3, c0 = (tid & 7) * 8;
            const int gk = k0 + kk;
            const bool kin = gk < K;
            const int64_t row = kin ? o.idx[gk] : 0;
            const double wk = kin ? o.w[gk] : 0.0;
            const double *xr = o.arx + row * (int64_t)o.n;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ci = c0 + u;
                const int64_t gi = m0 + ci;
                const int gj = n0 + ci;
                double ya = 0.0, yb = 0.0;
                if (kin && gi < M) ya = ((xr[gi] - o.xold[gi]) / o.sigma) * wk;  // artmp.T @ diag(w)
                if (kin && gj < N) yb = (xr[gj] - o.xold[gj]) / o.sigma;
                As[ci * LDA + kk] = ya;
                Bs[kk * LDB + ci] = yb;
            }
        }
        __syncthreads();
        // ---- 8 k-steps of 4: 2 A fragments x 2 B fragments -> 4 MFMAs per step ----
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const int kq = ks + (lane >> 4);
            const double a0 = As[(wm + (lane & 15)) * LDA + kq];
            const double a1 = As[(wm + 16 + (lane & 15)) * LDA + kq];
            const double b0 = Bs[kq * LDB + wn + (lane & 15)];
            const double b1 = Bs[kq * LDB + wn + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D element (row = (lane>>4) + 4*reg, col = lane&15) of each 16x16 tile ----
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wm + ti * 16 + (lane >> 4) + 4 * r;
                const int gj = n0 + wn + tj * 16 + (lane & 15);
                if (gi >= M || gj >= N) continue;
                const double g = acc[ti][tj][r];
                if (MODE == 0) {
                    const SampleOp &o = (const SampleOp &)op;
                    o.arx[gi * (int64_t)o.n + gj] = o.xmean[gj] + o.sigma * g;  // xmean + sigma * dot(B, D*z)
                } else {
                    const RankMuOp &o = (const RankMuOp &)op;
                    double *cp = o.C + gi * (int64_t)o.n + gj;
                    const double cold = *cp;
                    double c = cold * o.decay;           // C *= 1 - c1 - cmu
                    c = c + o.cmu * g;                   // C += cmu * A^T diag(w) A
                    c = c + o.c1 * (o.pc[gi] * o.pc[gj]);  // C += c1 * outer(pc, pc)
                    c = c + o.tmpc * cold;               // C += tmp  (tmp = c1*cc*(2-cc)*C_old, or 0)
                    *cp = c;
                }
            }
        }
    }
}

// xmean[e] = sum_k w[k] * arx[idx[k]][e]   (one workgroup per 64 columns; 16 k-slices of 64 lanes each,
// 4 rows in flight per thread, fixed combination order => reproducible)
constexpr int kRecSlices = 16;
__global__ __launch_bounds__(64 * kRecSlices) void cma_recombine_kernel(const double *__restrict__ arx,
                                                                        const int64_t *__restrict__ idx,
                                                                        const double *__restrict__ w, int mu, int n,
                                                                        double *__restrict__ xmean) {
    __shared__ double part[kRecSlices][64];
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 64 + lane;
    const int slice = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < n) {
        for (int k0 = slice; k0 < mu; k0 += 4 * kRecSlices) {
            double v[4], ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * kRecSlices;
                const bool in = k < mu;
                ww[u] = in ? w[k] : 0.0;
                v[u] = in ? arx[idx[k] * (int64_t)n + col] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += ww[u] * v[u];
        }
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && col < n) {
        double s = part[0][lane];
#pragma unroll
        for (int k = 1; k < kRecSlices; ++k) s += part[k][lane];
        xmean[col] = s;
    }
}

// Z[i][e] ~ N(0,1): Box-Muller on the two 53-bit uniforms of a call, half 0 -> cos, half 1 -> sin
// (oracle/streams.py PhiloxStream.cma_normals)
__global__ __launch_bounds__(256) void cma_normals_kernel(double *__restrict__ Z, int64_t P, int n, int64_t row0,
                                                          uint32_t gen, uint32_t k0, uint32_t k1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= P) return;
    const uint32_t grow = (uint32_t)(row0 + row);
    const uint32_t lpr = (uint32_t)lanes_per_row(n);  // same element -> (slot, half) layout as the row kernels
    for (int e = lane; e < n; e += kWave) {
        const uint32_t q = (uint32_t)e / lpr, l = (uint32_t)e & (lpr - 1u);
        const U4 w = philox4x32_10((q >> 1) * lpr + l, grow, gen, kPurposeCmaNormal, k0, k1);
        const double d0 = u53(w.x, w.y), d1 = u53(w.z, w.w);
        const double rad = sqrt(-2.0 * log(1.0 - d0));
        const double ang = 6.283185307179586 * d1;
        Z[row * (int64_t)n + e] = (q & 1u) ? rad * sin(ang) : rad * cos(ang);
    }
}

// C = triu(C) + triu(C,1).T
__global__ __launch_bounds__(256) void symmetrize_upper_kernel(double *__restrict__ C, int n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * n) return;
    const int i = (int)(t / n), j = (int)(t % n);
    if (i > j) C[t] = C[(int64_t)j * n + i];
}

}  // namespace

extern "C" int sx_cmaes_sample(const double *xmean, double sigma, const double *B, const double *D, const double *Z,
                               double *arx, int64_t P, int n, void *stream) {
    SX_REQUIRE(xmean && B && D && Z && arx && P >= 1 && n >= 1, "sx_cmaes_sample: bad arguments");
    SampleOp op{Z, B, D, xmean, arx, sigma, P, n};
    dim3 grid((unsigned)((n + BN - 1) / BN), (unsigned)((P + BM - 1) / BM));
    hipLaunchKernelGGL((cma_gemm_kernel<0, SampleOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_cmaes_rank_mu(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold,
                                double sigma, const double *pc, double c1, double cmu, double tmp_coef, double *C,
                                int n, void *stream) {
    SX_REQUIRE(arx && idx && w && xold && pc && C && mu >= 1 && n >= 1, "sx_cmaes_rank_mu: bad arguments");
    RankMuOp op{arx, idx, w, xold, pc, C, sigma, 1.0 - c1 - cmu, cmu, c1, tmp_coef, mu, n};
    dim3 grid((unsigned)((n + BN - 1) / BN), (unsigned)((n + BM - 1) / BM));
    hipLaunchKernelGGL((cma_gemm_kernel<1, RankMuOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_cmaes_recombine(const double *arx, const int64_t *idx, const double *w, int mu, int n,
                                  double *xmean, void *stream) {
    SX_REQUIRE(arx && idx && w && xmean && mu >= 1 && n >= 1, "sx_cmaes_recombine: bad arguments");
    hipLaunchKernelGGL(cma_recombine_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * kRecSlices), 0, (hipStream_t)stream, arx, idx,
                       w, mu, n, xmean);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_cmaes_normals(double *Z, int64_t P, int n, int64_t row0, uint32_t gen, uint32_t key0, uint32_t key1,
                                void *stream) {
    SX_REQUIRE(Z && P >= 1 && n >= 1, "sx_cmaes_normals: bad arguments");
    hipLaunchKernelGGL(cma_normals_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Z, P, n,
                       row0, gen, key0, key1);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_symmetrize_upper(double *C, int n, void *stream) {
    SX_REQUIRE(C && n >= 1, "sx_symmetrize_upper: bad arguments");
    const int64_t total = (int64_t)n * n;
    hipLaunchKernelGGL(symmetrize_upper_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, C, n);
    SX_LAUNCH_CHECK();
    return 0;
}
