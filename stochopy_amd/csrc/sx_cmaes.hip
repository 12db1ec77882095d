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
// Workgroup tiles 64x64 / 32x64 / 32x32 (chosen so the small CMA-ES shapes still give >= 256 workgroups),
// four waves in a 2x2 grid, K chunk 32 staged through LDS, the next three chunks already on their way in registers.
#include <cstdlib>
#include <type_traits>
#ifndef SX_GEMM_XCD_SWIZZLE
#define SX_GEMM_XCD_SWIZZLE 1  // (0: blockIdx.x = column tile, blockIdx.y = row tile as launched -- A/B builds)
#endif
#include "sx_device.hpp"
#include "sx_host.hpp"

using namespace sx;

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile double lds_cvd;  // a volatile read that stays a DS instruction

constexpr int KC = 32;        // K chunk
constexpr int LDA = KC + 2;   // k-major tiles [rows][LDA]: row stride 34 doubles -> conflict-free ds_read_b64 of a column slab
constexpr int kGemmThreads = 256;
constexpr int kGemmAhead = 3;  // operand chunks in flight per thread

struct SampleOp {  // arx = xmean + sigma * (Z o D) * B^T
    const double *Z;      // (P,n)
    const double *Bm;     // (n,n) eigenvectors, row-major
    const double *D;      // (n)
    const double *xmean;  // (n)
    double *arx;          // (P,n)
    double sigma;
    const double *sigma_p;  // when set: the step size lives on the device (device-resident loop)
    int64_t P;
    int n;
};

struct RankMuOp {  // C = (1-c1-cmu)*C + cmu * Y^T diag(w) Y + c1 * pc pc^T + tmpc * C_old
    const double *Y;      // (mu,n)  Y[k] = (arx[idx[k]] - xold)/sigma  (cma_y_kernel)
    const double *w;      // (mu)
    const double *pc;     // (n)
    double *C;            // (n,n) in place
    double decay, cmu, c1, tmpc;
    const double *tmpc_p;  // when set: tmp coefficient on the device (0 when `cond` held, else c1*cc*(2-cc))
    int mu, n;
    // round 4: only the upper triangle is ever read back (cmaes/_cmaes.py:303 mirrors triu(C); :290-295 is elementwise), so
    // only the tiles on and above the diagonal are formed -- `tri` tile rows -- and the chip is filled by splitting K in two
    // instead: blockIdx.z = which half of the mu terms, raw products to part[z] (n x n each), cma_cov_finish_kernel adds the
    // halves in a fixed order, applies the update and mirrors.  part == NULL: the full matrix, updated in place (as before).
    double *part;
    int tri;     // tiles per side when the grid enumerates the upper triangle (blockIdx.x = tile number), else 0
    int ksplit;  // 1 or 2
};

// Y[k][:] = (arx[idx[k]][:] - xold) / sigma   (cmaes/_cmaes.py:290), once per generation
__global__ __launch_bounds__(256) void cma_y_kernel(const double *__restrict__ arx, const int64_t *__restrict__ idx,
                                                    const double *__restrict__ xold, double sigma,
                                                    const double *__restrict__ sigma_p, int mu, int n,
                                                    double *__restrict__ Y) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)mu * n) return;
    if (sigma_p) sigma = *sigma_p;
    const int k = (int)(t / n), e = (int)(t % n);
    Y[t] = (arx[idx[k] * (int64_t)n + e] - xold[e]) / sigma;
}

// MODE 0: M = P rows, N = n, K = n.   MODE 1: M = N = n, K = mu.
// Workgroup tile BM x BN, four waves in a 2x2 grid, wave tile (BM/2) x (BN/2) of 16x16 MFMA tiles.
// The global loads of chunks k+1 .. k+3 are in registers or in flight while the MFMAs of chunk k run.
template <int MODE, int BM, int BN, class Op>
__global__ __launch_bounds__(kGemmThreads) void cma_gemm_kernel(const Op op) {
    // LDS images follow the way the operands lie in memory, so staging is a contiguous copy for both of them:
    //   MODE 0  Z and B are rows along k      -> As[BM][LDA], Bs[BN][LDA]        (fragment reads: row stride 34 doubles)
    //   MODE 1  Y is rows along the tile axes -> As[KC][BM + 16], Bs[KC][BN + 16] (fragment reads: k stride = 16 mod 32 doubles)
    // Both give conflict-free ds_read_b64 fragments (32-lane halves, 64 banks) and at most 2-way staging stores.
    constexpr int LDM = BM + 16, LDN = BN + 16;
    constexpr int TM = BM / 32, TN = BN / 32;  // MFMA tiles per wave
    constexpr int NA = BM * KC / kGemmThreads, NB = BN * KC / kGemmThreads;  // staged elements per thread
    constexpr int ASZ = MODE == 0 ? BM * LDA : KC * LDM, BSZ = MODE == 0 ? BN * LDA : KC * LDN;
    // two LDS images where they fit the 64 KB of static LDS: chunk c+1 is staged while chunk c is multiplied, one barrier per chunk
    constexpr int NBUF = 2 * (ASZ + BSZ) * 8 <= 65536 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) double As_[NBUF * ASZ];
    __shared__ __attribute__((aligned(16))) double Bs_[NBUF * BSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    int64_t m0 = (int64_t)blockIdx.y * BM;
    int n0 = blockIdx.x * BN;
#if SX_GEMM_XCD_SWIZZLE
    if (MODE == 0) {
        // Workgroups go to the 8 XCDs round-robin by their linear number.  Numbered with the COLUMN tile fastest an XCD meets
        // every candidate row (all of Z: 4 MB at C4, the size of its L2) and two column tiles of B D; numbered with the ROW tile
        // fastest it meets an eighth of Z (0.5 MB) and all of B D (2 MB), which its L2 holds.  (Isolated launches 17.05 -> 16.55 us, in
        // situ no difference: profiles/r4_gemm_xcd_numbering.txt.)
        const unsigned id = blockIdx.x + gridDim.x * blockIdx.y;
        m0 = (int64_t)(id % gridDim.y) * BM;
        n0 = (int)(id / gridDim.y) * BN;
    }
#endif
    int64_t M;
    int N, K;
    int kbase = 0;  // MODE 1 with split K: this workgroup's first term
    if (MODE == 0) {
        const SampleOp &o = (const SampleOp &)op;
        M = o.P, N = o.n, K = o.n;
    } else {
        const RankMuOp &o = (const RankMuOp &)op;
        M = o.n, N = o.n, K = o.mu;
        if (o.tri > 0) {  // tile number -> (row block bi <= column block bj), row by row of the upper triangle
            int t = (int)blockIdx.x, bi = 0, len = o.tri;
            while (t >= len) t -= len, ++bi, --len;  // (uniform, at most `tri` steps)
            m0 = (int64_t)bi * BM;
            n0 = (bi + t) * BN;
        }
        if (o.ksplit > 1) {  // halves of whole K chunks
            const int chunks = (K + KC - 1) / KC, first = (int)blockIdx.z * (chunks / 2) * KC;
            const int last = blockIdx.z == 0 ? (chunks / 2) * KC : K;
            kbase = first, K = last - first;
        }
    }
    v4d acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // kGemmAhead chunks of operands live in registers: a workgroup is one wave per SIMD and the small CMA-ES shapes put one or
    // two workgroups on a CU, so the only thing that hides the ~1 us of a global load is having several chunks in flight.
    // fetch() only LOADS (nothing that needs the value): every use -- the D and w factors, the zero fill of the ragged
    // edge -- waits until stage(), three chunks later.  A workgroup whose tile and K range lie inside the operands (FULL)
    // runs without a single guard: no exec-masked branches, so the compiler's s_waitcnt vmcnt(N) can leave the younger
    // chunks in flight (with the guards in the way it waits for vmcnt(0) at every stage and the depth collapses to one).
    double ra[kGemmAhead][NA], rb[kGemmAhead][NB];
    double rs[kGemmAhead][2];  // MODE 0: D[k] of the thread's two k; MODE 1: w[k] of the thread's two k rows
    const int nchunk = (K + KC - 1) / KC;
    // thread -> staged elements: 16 consecutive lanes take the 16 pieces of 16 B that make up one 256-byte row of a chunk
    // (MODE 0: 32 k of one candidate / eigenvector row; MODE 1: 32 columns of one Y row), so a wave's load is four whole
    // 256-byte runs -- the address unit (TA) was the busiest block of this kernel when lanes gathered 16-byte pieces from
    // eight rows each -- and its ds_write_b128 (served 8 consecutive lanes at a time) fills 8 consecutive 16-byte slots.
    //   MODE 0  row = 16 q + tid/16 (q < rows/16),            k = 2 (tid%16) + {0,1}
    //   MODE 1  k   = 16 q + tid/16 (q < 2),   col = 32 c + 2 (tid%16) + {0,1} (c < cols/32)
    const int t16 = tid >> 4, p2 = 2 * (tid & 15);
    auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        auto fetch = [&](const int s, const int k0) {
            if (MODE == 0) {
                const SampleOp &o = (const SampleOp &)op;
                int gk[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    gk[u] = FULL || k0 + p2 + u < K ? k0 + p2 + u : K - 1;
                    rs[s][u] = o.D[gk[u]];
                }
#pragma unroll
                for (int q = 0; q < NA / 2; ++q) {
                    const int64_t gi = m0 + q * 16 + t16;
                    const double *zr = o.Z + (FULL || gi < M ? gi : M - 1) * (int64_t)o.n;
#pragma unroll
                    for (int u = 0; u < 2; ++u) ra[s][q * 2 + u] = zr[gk[u]];
                }
#pragma unroll
                for (int q = 0; q < NB / 2; ++q) {
                    const int gj = n0 + q * 16 + t16;
                    const double *br = o.Bm + (int64_t)(FULL || gj < N ? gj : N - 1) * o.n;
#pragma unroll
                    for (int u = 0; u < 2; ++u) rb[s][q * 2 + u] = br[gk[u]];
                }
            } else {
                const RankMuOp &o = (const RankMuOp &)op;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int gk = kbase + (FULL || k0 + q * 16 + t16 < K ? k0 + q * 16 + t16 : K - 1);
                    const double *yr = o.Y + (int64_t)gk * o.n;
                    rs[s][q] = o.w[gk];
#pragma unroll
                    for (int c = 0; c < BM / 32; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int64_t gi = m0 + c * 32 + p2 + u;
                            ra[s][(q * (BM / 32) + c) * 2 + u] = yr[FULL || gi < M ? gi : M - 1];
                        }
#pragma unroll
                    for (int c = 0; c < BN / 32; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int gj = n0 + c * 32 + p2 + u;
                            rb[s][(q * (BN / 32) + c) * 2 + u] = yr[FULL || gj < N ? gj : N - 1];
                        }
                }
            }
        };
        auto stage = [&](const int s, const int k0, const int buf) {
            double *As = As_ + buf * ASZ, *Bs = Bs_ + buf * BSZ;
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < NA / 2; ++q) {
                    const bool rin = FULL || m0 + q * 16 + t16 < M;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {  // D * z (:234)
                        const double v = rs[s][u] * ra[s][q * 2 + u];
                        As[(q * 16 + t16) * LDA + p2 + u] = (FULL || (rin && k0 + p2 + u < K)) ? v : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < NB / 2; ++q) {
                    const bool rin = FULL || n0 + q * 16 + t16 < N;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        Bs[(q * 16 + t16) * LDA + p2 + u] = (FULL || (rin && k0 + p2 + u < K)) ? rb[s][q * 2 + u] : 0.0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const bool kin = FULL || k0 + q * 16 + t16 < K;
#pragma unroll
                    for (int c = 0; c < BM / 32; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {  // artmp.T @ diag(w)
                            const double v = ra[s][(q * (BM / 32) + c) * 2 + u] * rs[s][q];
                            As[(q * 16 + t16) * LDM + c * 32 + p2 + u] = (FULL || (kin && m0 + c * 32 + p2 + u < M)) ? v : 0.0;
                        }
#pragma unroll
                    for (int c = 0; c < BN / 32; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            Bs[(q * 16 + t16) * LDN + c * 32 + p2 + u] =
                                (FULL || (kin && n0 + c * 32 + p2 + u < N)) ? rb[s][(q * (BN / 32) + c) * 2 + u] : 0.0;
                }
            }
        };
        auto multiply = [&](const int buf) {
            const double *As = As_ + buf * ASZ, *Bs = Bs_ + buf * BSZ;
            // All fragment reads of the chunk are issued before its first MFMA: a wave is one dependent chain per accumulator,
            // so a read placed between two MFMAs puts the LDS latency on that chain 8 times per chunk.
            // volatile: one ds_read_b64 per fragment (2 LDS cycles, conflict-free in both layouts); left alone the compiler
            // pairs the reads of two k steps into ds_read2_b64, which the LDS serves at half that rate.
            // (with 2x2 tiles per wave there are four chains to interleave and the registers are better spent elsewhere: G = 1)
            constexpr int G = TM * TN == 1 ? KC / 4 : 1;  // k steps whose fragments are read together
            double af[G][TM], bf[G][TN];
#pragma unroll
            for (int x0 = 0; x0 < KC / 4; x0 += G) {
#pragma unroll
            for (int x = 0; x < G; ++x) {
                const int kq = 4 * (x0 + x) + (lane >> 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[x][i] = *(lds_cvd *)(MODE == 0 ? &As[(wm + i * 16 + (lane & 15)) * LDA + kq]
                                                      : &As[kq * LDM + wm + i * 16 + (lane & 15)]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[x][j] = *(lds_cvd *)(MODE == 0 ? &Bs[(wn + j * 16 + (lane & 15)) * LDA + kq]
                                                      : &Bs[kq * LDN + wn + j * 16 + (lane & 15)]);
            }
#pragma unroll
            for (int x = 0; x < G; ++x)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][i], bf[x][j], acc[i][j], 0, 0, 0);
            }
        };
        // chunk c lives in register set c % kGemmAhead; every loop below is unrolled so that the set is a compile-time index
#pragma unroll
        for (int s = 0; s < kGemmAhead; ++s)
            if (s < nchunk) fetch(s, s * KC);
        int c0 = 0;
        if (NBUF == 1) {
            auto step = [&](const int s, const int c, const bool more) {
                stage(s, c * KC, 0);
                __syncthreads();
                if (more) fetch(s, (c + kGemmAhead) * KC);  // in flight for the next kGemmAhead chunks
                multiply(0);
                __syncthreads();
            };
            for (; c0 + 2 * kGemmAhead <= nchunk; c0 += kGemmAhead) {  // steady state: every step also fetches, no conditions
#pragma unroll
                for (int s = 0; s < kGemmAhead; ++s) step(s, c0 + s, true);
            }
#pragma unroll
            for (int t = 0; t < 2 * kGemmAhead - 1; ++t) {  // the last kGemmAhead .. 2*kGemmAhead-1 chunks
                const int c = c0 + t;
                if (c < nchunk) step(t % kGemmAhead, c, c + kGemmAhead < nchunk);
            }
        } else {
            auto step = [&](const int s, const int c, const bool next, const bool more) {
                constexpr int A = kGemmAhead;
                if (next) stage((s + 1) % A, (c + 1) * KC, (c + 1) & 1);  // the image the previous step multiplied from
                if (more) fetch((s + 1) % A, (c + 1 + A) * KC);
                multiply(c & 1);
                __syncthreads();
            };
            stage(0, 0, 0);
            if (kGemmAhead < nchunk) fetch(0, kGemmAhead * KC);
            __syncthreads();
            for (; c0 + 2 * kGemmAhead + 1 <= nchunk; c0 += kGemmAhead) {
#pragma unroll
                for (int s = 0; s < kGemmAhead; ++s) step(s, c0 + s, true, true);
            }
#pragma unroll
            for (int t = 0; t < 2 * kGemmAhead; ++t) {
                const int c = c0 + t;
                if (c < nchunk) step(t % kGemmAhead, c, c + 1 < nchunk, c + 1 + kGemmAhead < nchunk);
            }
        }
    };
    if (m0 + BM <= M && n0 + BN <= N && K % KC == 0)
        run(std::true_type{});
    else
        run(std::false_type{});
    // ---- epilogue: C/D element (row = (lane>>4) + 4*reg, col = lane&15) of each 16x16 tile ----
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gi = m0 + wm + ti * 16 + (lane >> 4) + 4 * r;
                const int gj = n0 + wn + tj * 16 + (lane & 15);
                if (gi >= M || gj >= N) continue;
                const double g = acc[ti][tj][r];
                if (MODE == 0) {
                    const SampleOp &o = (const SampleOp &)op;
                    const double sg = o.sigma_p ? *o.sigma_p : o.sigma;
                    o.arx[gi * (int64_t)o.n + gj] = o.xmean[gj] + sg * g;  // xmean + sigma * dot(B, D*z)
                } else {
                    const RankMuOp &o = (const RankMuOp &)op;
                    if (o.part != nullptr) {  // raw products of this half of K; cma_cov_finish_kernel does the rest
                        o.part[((int64_t)blockIdx.z * o.n + gi) * (int64_t)o.n + gj] = g;
                        continue;
                    }
                    const double tmpc = o.tmpc_p ? *o.tmpc_p : o.tmpc;
                    double *cp = o.C + gi * (int64_t)o.n + gj;
                    const double cold = *cp;
                    double c = cold * o.decay;             // C *= 1 - c1 - cmu
                    c = c + o.cmu * g;                     // C += cmu * A^T diag(w) A
                    c = c + o.c1 * (o.pc[gi] * o.pc[gj]);  // C += c1 * outer(pc, pc)
                    c = c + tmpc * cold;                   // C += tmp  (tmp = c1*cc*(2-cc)*C_old, or 0)
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
    // elements (2k)*lpr + l and (2k+1)*lpr + l are the cosine and the sine half of ONE call (slot k*lpr + l): a lane
    // takes both, so the generator, the logarithm, the square root and the angle reduction run once per pair
    const int npair = ((n + 2 * (int)lpr - 1) / (2 * (int)lpr)) * (int)lpr;
    double *zr = Z + row * (int64_t)n;
    for (int j = lane; j < npair; j += kWave) {
        const uint32_t k = (uint32_t)j / lpr, l = (uint32_t)j & (lpr - 1u);
        const int e0 = (int)(2u * k * lpr + l), e1 = e0 + (int)lpr;
        if (e0 >= n) continue;
        const U4 w = philox4x32_10(k * lpr + l, grow, gen, kPurposeCmaNormal, k0, k1);
        const double d0 = u53(w.x, w.y), d1 = u53(w.z, w.w);
        const double rad = sqrt(-2.0 * log(1.0 - d0));
        const double ang = 6.283185307179586 * d1;
        double sn, cs;
        sincos(ang, &sn, &cs);
        zr[e0] = rad * cs;
        if (e1 < n) zr[e1] = rad * sn;
    }
}

// C = triu(C) + triu(C,1).T
__global__ __launch_bounds__(256) void symmetrize_upper_kernel(double *__restrict__ C, int n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * n) return;
    const int i = (int)(t / n), j = (int)(t % n);
    if (i > j) C[t] = C[(int64_t)j * n + i];
}

// the covariance update from the two half-sums of cma_gemm_kernel<1> (RankMuOp.part), upper triangle, mirrored:
// C_ij = C_ij (1-c1-cmu) + cmu (G0_ij + G1_ij) + c1 pc_i pc_j + tmp C_ij(old)  -- the operations of :291-295 in their order
__global__ __launch_bounds__(256) void cma_cov_finish_kernel(const RankMuOp o, int nparts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = o.n;
    if (t >= (int64_t)n * n) return;
    const int i = (int)(t / n), j = (int)(t % n);
    if (i > j) return;
    const double g = nparts > 1 ? o.part[t] + o.part[(int64_t)n * n + t] : o.part[t];
    const double tmpc = o.tmpc_p ? *o.tmpc_p : o.tmpc;
    const double cold = o.C[t];
    double c = cold * o.decay;
    c = c + o.cmu * g;
    c = c + o.c1 * (o.pc[i] * o.pc[j]);
    c = c + tmpc * cold;
    o.C[t] = c;
    if (i != j) o.C[(int64_t)j * n + i] = c;
}

}  // namespace

// tile choice: enough workgroups to cover the 256 CUs on the (small) CMA-ES shapes
namespace sx {
int cma_sample_launch(const double *xmean, double sigma, const double *sigma_p, const double *B, const double *D,
                      const double *Z, double *arx, int64_t P, int n, void *stream);
int cma_rank_mu_launch(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold, double sigma,
                       const double *sigma_p, const double *pc, double c1, double cmu, double tmp_coef,
                       const double *tmp_coef_p, double *C, double *ws_y, int n, void *stream, double *split_ws = nullptr,
                       int *mirrored = nullptr);
}  // namespace sx

extern "C" int sx_cmaes_sample(const double *xmean, double sigma, const double *B, const double *D, const double *Z,
                               double *arx, int64_t P, int n, void *stream) {
    return sx::cma_sample_launch(xmean, sigma, nullptr, B, D, Z, arx, P, n, stream);
}

int sx::cma_sample_launch(const double *xmean, double sigma, const double *sigma_p, const double *B, const double *D,
                          const double *Z, double *arx, int64_t P, int n, void *stream) {
    SX_REQUIRE(xmean && B && D && Z && arx && P >= 1 && n >= 1, "sx_cmaes_sample: bad arguments");
    SampleOp op{Z, B, D, xmean, arx, sigma, sigma_p, P, n};
    const int64_t big = ((P + 63) / 64) * ((n + 63) / 64);
    if (big >= 512) {
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((P + 63) / 64));
        hipLaunchKernelGGL((cma_gemm_kernel<0, 64, 64, SampleOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    } else if (big >= 256) {
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((P + 31) / 32));
        hipLaunchKernelGGL((cma_gemm_kernel<0, 32, 64, SampleOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    } else {  // small problems: more, smaller workgroups (2+ waves per SIMD hide the per-chunk latency)
        dim3 grid((unsigned)((n + 31) / 32), (unsigned)((P + 31) / 32));
        hipLaunchKernelGGL((cma_gemm_kernel<0, 32, 32, SampleOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    }
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_cmaes_rank_mu(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold,
                                double sigma, const double *pc, double c1, double cmu, double tmp_coef, double *C,
                                double *ws_y, int n, void *stream) {
    return sx::cma_rank_mu_launch(arx, idx, w, mu, xold, sigma, nullptr, pc, c1, cmu, tmp_coef, nullptr, C, ws_y, n, stream);
}

// split_ws: optional DEVICE scratch of 2 n^2 doubles -> the upper-triangle / split-K form (the result is then symmetric:
// *mirrored = 1); without it the whole matrix is updated in place by the contraction kernel itself.
int sx::cma_rank_mu_launch(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold, double sigma,
                           const double *sigma_p, const double *pc, double c1, double cmu, double tmp_coef,
                           const double *tmp_coef_p, double *C, double *ws_y, int n, void *stream, double *split_ws,
                           int *mirrored) {
    if (mirrored) *mirrored = 0;
    SX_REQUIRE(arx && idx && w && xold && pc && C && ws_y && mu >= 1 && n >= 1, "sx_cmaes_rank_mu: bad arguments");
    const int64_t total = (int64_t)mu * n;
    hipLaunchKernelGGL(cma_y_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, arx, idx,
                       xold, sigma, sigma_p, mu, n, ws_y);
    SX_LAUNCH_CHECK();
    RankMuOp op{ws_y, w, pc, C, 1.0 - c1 - cmu, cmu, c1, tmp_coef, tmp_coef_p, mu, n, nullptr, 0, 1};
    const int64_t big = ((int64_t)(n + 63) / 64) * ((n + 63) / 64);
    static const bool tri_ok = !(getenv("SX_CMA_TRI") && getenv("SX_CMA_TRI")[0] == '0');  // (A/B switch)
    if (split_ws != nullptr && tri_ok && n >= 128 && mu >= 4 * KC) {
        op.part = split_ws, op.ksplit = 2;
        if (big >= 1024) {  // (twice the full form's bound: the triangle has half the tiles, the split doubles them again)
            op.tri = (n + 63) / 64;
            hipLaunchKernelGGL((cma_gemm_kernel<1, 64, 64, RankMuOp>), dim3((unsigned)(op.tri * (op.tri + 1) / 2), 1, 2),
                               dim3(kGemmThreads), 0, (hipStream_t)stream, op);
        } else {
            op.tri = (n + 31) / 32;
            hipLaunchKernelGGL((cma_gemm_kernel<1, 32, 32, RankMuOp>), dim3((unsigned)(op.tri * (op.tri + 1) / 2), 1, 2),
                               dim3(kGemmThreads), 0, (hipStream_t)stream, op);
        }
        SX_LAUNCH_CHECK();
        hipLaunchKernelGGL(cma_cov_finish_kernel, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, op, 2);
        SX_LAUNCH_CHECK();
        if (mirrored) *mirrored = 1;
        return 0;
    }
    if (big >= 512) {
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64));
        hipLaunchKernelGGL((cma_gemm_kernel<1, 64, 64, RankMuOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    } else {
        dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32));
        hipLaunchKernelGGL((cma_gemm_kernel<1, 32, 32, RankMuOp>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, op);
    }
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

// ---------------------------------------------------------------------------
// VD-CMA sampling (stochopy/optimize/vdcma/_vdcma.py:236-248): with the covariance model D (I + v v^T) D a
// candidate costs O(n):  y = d o (z + (sqrt(1 + |v|^2) - 1) (z . vn) vn),  x = xmean + sigma * y.
// One wavefront per candidate; rows 0 and 1 of the generation are replaced by +dy / -dy when the mean-shift
// injection is on (:241-247).  Both y (needed by the host's moment sums) and x are written.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void vd_sample_kernel(const double *__restrict__ Z, int64_t P, int n, int64_t row0,
                                                        const double *__restrict__ dvec, const double *__restrict__ vn,
                                                        double coef, const double *__restrict__ xmean, double sigma,
                                                        const double *__restrict__ dy, double *__restrict__ ary,
                                                        double *__restrict__ arx, const sx_cma_state *st) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= P) return;
    if (st != nullptr) {  // device-resident loop: step size, model coefficient and injection flag live on the device
        if (st->done) return;
        sigma = st->sigma;
        coef = st->reserved[4];
        if (st->reserved[3] == 0.0) dy = nullptr;
    }
    const double *z = Z + row * (int64_t)n;
    double t = 0.0;
    // 8 row loads per lane in flight (the row is streamed twice: the second pass hits L2)
    int e = lane;
    for (; e + 7 * kWave < n; e += 8 * kWave) {
        double zz[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) zz[u] = z[e + u * kWave], vv[u] = vn[e + u * kWave];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += zz[u] * vv[u];
    }
    for (; e < n; e += kWave) t += z[e] * vn[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, kWave);
    const int64_t grow = row0 + row;
    const bool inj = dy != nullptr && grow < 2;
    const double sgn = grow == 0 ? 1.0 : -1.0;
    double *yo = ary + row * (int64_t)n, *xo = arx + row * (int64_t)n;
    e = lane;
    for (; e + 3 * kWave < n; e += 4 * kWave) {
        double zz[4], vv[4], dd[4], xm[4], dj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = e + u * kWave;
            zz[u] = z[c], vv[u] = vn[c], dd[u] = dvec[c], xm[u] = xmean[c], dj[u] = inj ? dy[c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = e + u * kWave;
            const double y = inj ? sgn * dj[u] : dd[u] * (zz[u] + coef * (t * vv[u]));
            yo[c] = y;
            xo[c] = xm[u] + sigma * y;
        }
    }
    for (; e < n; e += kWave) {
        const double y = inj ? sgn * dy[e] : dvec[e] * (z[e] + coef * (t * vn[e]));
        yo[e] = y;
        xo[e] = xmean[e] + sigma * y;
    }
}
}  // namespace

extern "C" int sx_vdcma_sample(const double *Z, int64_t P, int n, int64_t row0, const double *dvec, const double *vn,
                               double coef, const double *xmean, double sigma, const double *dy, double *ary,
                               double *arx, void *stream) {
    SX_REQUIRE(Z && dvec && vn && xmean && ary && arx && P >= 1 && n >= 1 && row0 >= 0, "sx_vdcma_sample: bad arguments");
    const int rows_per_block = 4;
    hipLaunchKernelGGL(vd_sample_kernel, dim3((unsigned)((P + rows_per_block - 1) / rows_per_block)),
                       dim3(rows_per_block * kWave), 0, (hipStream_t)stream, Z, P, n, row0, dvec, vn, coef, xmean, sigma,
                       dy, ary, arx, (const sx_cma_state *)nullptr);
    SX_LAUNCH_CHECK();
    return 0;
}

namespace sx {
// the same with sigma / coefficient / injection flag read from the device state (sx_vdcma_generation)
int vd_sample_launch(const double *Z, int64_t P, int n, const double *dvec, const double *vn, const double *xmean,
                     const double *dy, double *ary, double *arx, const sx_cma_state *st, void *stream, int64_t row0) {
    const int rows_per_block = 4;
    hipLaunchKernelGGL(vd_sample_kernel, dim3((unsigned)((P + rows_per_block - 1) / rows_per_block)),
                       dim3(rows_per_block * kWave), 0, (hipStream_t)stream, Z, P, n, row0, dvec, vn, 0.0, xmean, 0.0,
                       dy, ary, arx, st);
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace sx

// ---------------------------------------------------------------------------
// VD-CMA moment sums (stochopy/optimize/vdcma/_vdcma.py:289-295 weighted mean of the selected candidates, :317 the
// weighted step w . y, :331-339 + :428-444 the rank-mu moments p, q under the model D (I + v v^T) D): everything
// that is O(mu n).  Only four n-vectors go back to the host per generation.
//   t_k   = (ary[idx_k] / dvec) . vn                                   (vd_t_kernel, one wavefront per selected row)
//   wx    = sum_k w_k arx[idx_k]            wy = sum_k w_k ary[idx_k]
//   p_mu  = sum_k w_k (y_k^2 - shrink * t_k * y_k * vn - 1)           with y_k = ary[idx_k] / dvec
//   q_mu  = sum_k w_k (t_k * y_k - 0.5 (t_k^2 + 1 + |v|^2) * vn)
// as 64 partial rows (k mod 64) per output, then a fixed-order finish.
// ---------------------------------------------------------------------------
namespace {
constexpr int kVdPart = 64;

__global__ __launch_bounds__(256) void vd_t_kernel(const double *__restrict__ ary, const int64_t *__restrict__ idx, int mu,
                                                   int n, const double *__restrict__ dvec, const double *__restrict__ vn,
                                                   double *__restrict__ tk) {
    const int lane = (int)(threadIdx.x & 63);
    const int k = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (k >= mu) return;
    const double *y = ary + idx[k] * (int64_t)n;
    double t = 0.0;
    int e = lane;
    for (; e + 7 * kWave < n; e += 8 * kWave) {  // 8 row loads per lane in flight
        double yy[8], dd[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) yy[u] = y[e + u * kWave], dd[u] = dvec[e + u * kWave], vv[u] = vn[e + u * kWave];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += (yy[u] / dd[u]) * vv[u];
    }
    for (; e < n; e += kWave) t += (y[e] / dvec[e]) * vn[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, kWave);
    if (lane == 0) tk[k] = t;
}

// grid: ceil(n/64) x 16; part[o][q][e], o = 0..3 (wx, wy, p, q), q = k mod 64
// W = 2: two adjacent columns per thread (16-byte loads: a wavefront takes 1 KB of a selected row, not 512 bytes -- the kernel is a
// gather of row pieces); n even.  Per column the same operations in the same order as W = 1.
template <int W>
__global__ __launch_bounds__(256) void vd_moments_partial_kernel(const double *__restrict__ arx, const double *__restrict__ ary,
                                                                 const int64_t *__restrict__ idx, const double *__restrict__ w,
                                                                 const double *__restrict__ tk, int mu, int n,
                                                                 const double *__restrict__ dvec, const double *__restrict__ vn,
                                                                 double norm_v2, double *__restrict__ part,
                                                                 const sx_cma_state *st, const int tk_by_row,
                                                                 const double *__restrict__ xmean) {
    // tk_by_row: tk holds t of EVERY candidate row (left by the wide candidates kernel, sx_wide.hip), not of the selected ones
    // arx == NULL (wide device loop, nobody else wants the candidates): x = xmean + sigma y is formed again here, by the
    // candidates kernel's own expression from the same operands -- the same bits as the stored row would hold
    if (st != nullptr) norm_v2 = st->reserved[1];
    const bool form_x = arx == nullptr;
    const double sigma = form_x ? st->sigma : 0.0;
    double xm0[W];
#pragma unroll
    for (int c = 0; c < W; ++c) xm0[c] = 0.0;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + tx) * W;
    const int q = blockIdx.y * 4 + ty;
    if (col >= n) return;
    const double shrink = norm_v2 / (1.0 + norm_v2);
    double dv[W], v[W], awx[W], awy[W], ap[W], aq[W];
#pragma unroll
    for (int c = 0; c < W; ++c) dv[c] = dvec[col + c], v[c] = vn[col + c], awx[c] = 0.0, awy[c] = 0.0, ap[c] = 0.0, aq[c] = 0.0;
    if (form_x) {
#pragma unroll
        for (int c = 0; c < W; ++c) xm0[c] = xmean[col + c];
    }
    // (the selected rows of this slice, B at a time: their loads are in flight together -- a slice is only mu / 64 rows long, one
    //  memory round trip each when taken one by one -- and enter the sums in the order of k, as before)
    constexpr int B = 4;
    for (int k0 = q; k0 < mu; k0 += B * kVdPart) {
        double ax[B][W], ay[B][W], wk[B], t[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int k = k0 + u * kVdPart;
            const bool on = k < mu;
            const int64_t r = on ? idx[k] : idx[k0];
            const int64_t row = r * (int64_t)n + col;
            wk[u] = on ? w[k] : 0.0, t[u] = tk[tk_by_row ? r : (on ? k : k0)];
            if constexpr (W >= 2) {
#pragma unroll
                for (int c = 0; c < W; c += 2) {
                    const double2 y2 = *reinterpret_cast<const double2 *>(ary + row + c);
                    ay[u][c] = y2.x, ay[u][c + 1] = y2.y;
                    if (form_x) {
                        ax[u][c] = xm0[c] + sigma * y2.x, ax[u][c + 1] = xm0[c + 1] + sigma * y2.y;
                    } else {
                        const double2 a2 = *reinterpret_cast<const double2 *>(arx + row + c);
                        ax[u][c] = a2.x, ax[u][c + 1] = a2.y;
                    }
                }
            } else {
                ay[u][0] = ary[row];
                ax[u][0] = form_x ? xm0[0] + sigma * ay[u][0] : arx[row];
            }
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (k0 + u * kVdPart >= mu) break;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                const double y = ay[u][c] / dv[c];
                awx[c] += wk[u] * ax[u][c];
                awy[c] += wk[u] * ay[u][c];
                ap[c] += wk[u] * (y * y - shrink * (t[u] * (y * v[c])) - 1.0);
                aq[c] += wk[u] * (t[u] * y - (0.5 * (t[u] * t[u] + 1.0 + norm_v2)) * v[c]);
            }
        }
    }
    const int64_t plane = (int64_t)kVdPart * n;
#pragma unroll
    for (int c = 0; c < W; ++c) {
        const int64_t o = (int64_t)q * n + col + c;
        part[o] = awx[c], part[plane + o] = awy[c], part[2 * plane + o] = ap[c], part[3 * plane + o] = aq[c];
    }
}

// grid: ceil(n/64) x 4 outputs; the 64 partial rows of a column are added in their order (four slices of 16, then the
// slices in order: the same value as one thread adding all 64 would NOT be bit-identical, and need not be -- the
// device loop and the host loop are compared to rounding)
__global__ __launch_bounds__(256) void vd_moments_finish_kernel(const double *__restrict__ part, int n, double *__restrict__ out) {
    __shared__ double sl[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx, o = blockIdx.y;
    const int64_t plane = (int64_t)kVdPart * n;
    double s = 0.0;
    if (e < n) {
        double v[kVdPart / 4];
#pragma unroll
        for (int q = 0; q < kVdPart / 4; ++q) v[q] = part[o * plane + (int64_t)(ty * (kVdPart / 4) + q) * n + e];
#pragma unroll
        for (int q = 0; q < kVdPart / 4; ++q) s += v[q];
    }
    sl[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < n) out[(int64_t)o * n + e] = ((sl[0][tx] + sl[1][tx]) + sl[2][tx]) + sl[3][tx];
}
}  // namespace

namespace sx {
int vd_moments_launch(const double *arx, const double *ary, const int64_t *idx, const double *w, int mu, int n,
                      const double *dvec, const double *vn, double norm_v2, const sx_cma_state *state, double *ws,
                      double *out, void *stream, const double *tk_rows, const double *xmean) {
    hipStream_t st = (hipStream_t)stream;
    SX_REQUIRE(arx != nullptr || (xmean != nullptr && state != nullptr), "vd_moments_launch: no candidates and nothing to form them from");
    double *tk = ws, *part = ws + ((mu + 7) / 8) * 8;
    if (tk_rows == nullptr)
        hipLaunchKernelGGL(vd_t_kernel, dim3((unsigned)((mu + 3) / 4)), dim3(256), 0, st, ary, idx, mu, n, dvec, vn, tk);
    // (four columns per thread were measured too: 33.9 / 148 us against 33.2 / 128.5 with two, 44.1 / 168.7 with one --
    //  n = 16 384, mu = 512 / 2048: profiles/r5_vdcma_moments_note.txt)
    const bool pairs = n % 2 == 0 && (((uintptr_t)arx | (uintptr_t)ary) & 15) == 0;  // (arx NULL: its bits are 0)
    if (pairs)
        hipLaunchKernelGGL(vd_moments_partial_kernel<2>, dim3((unsigned)((n / 2 + 63) / 64), kVdPart / 4), dim3(256), 0, st, arx, ary,
                           idx, w, tk_rows ? tk_rows : (const double *)tk, mu, n, dvec, vn, norm_v2, part, state, tk_rows ? 1 : 0, xmean);
    else
        hipLaunchKernelGGL(vd_moments_partial_kernel<1>, dim3((unsigned)((n + 63) / 64), kVdPart / 4), dim3(256), 0, st, arx, ary, idx,
                           w, tk_rows ? tk_rows : (const double *)tk, mu, n, dvec, vn, norm_v2, part, state, tk_rows ? 1 : 0, xmean);
    hipLaunchKernelGGL(vd_moments_finish_kernel, dim3((unsigned)((n + 63) / 64), 4), dim3(256), 0, st, part, n, out);
    SX_LAUNCH_CHECK();
    return 0;
}
}  // namespace sx

extern "C" int sx_vdcma_moments(const double *arx, const double *ary, const int64_t *idx, const double *w, int mu, int n,
                                const double *dvec, const double *vn, double norm_v2, double *ws, double *out, void *stream) {
    SX_REQUIRE(arx && ary && idx && w && dvec && vn && ws && out && mu >= 1 && n >= 1, "sx_vdcma_moments: bad arguments");
    return sx::vd_moments_launch(arx, ary, idx, w, mu, n, dvec, vn, norm_v2, nullptr, ws, out, stream, nullptr, nullptr);
}
