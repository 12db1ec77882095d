// Generations around a CALLER-SUPPLIED objective (backend hook contract: a callable that maps the
// (P, n) device array to (P,) fitness values).  The objective cannot be fused, so a generation is
//   propose / move kernel  ->  caller's objective on the device array  ->  selection kernel  ->  sx_select_finalize
// with the same draws, arithmetic and record layout as the fused kernels: given bit-identical fitness values the
// run is the fused run, bit for bit.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/de/_de.py:314-351       de_sync up to the candidates U (mutation, crossover, Random)
//   stochopy/optimize/cpso/_cpso.py:324-329   mutation (velocity / position) + cpso/_constraints.py:4-53
//   stochopy/optimize/_common.py:123-130      selection_sync after `candfun = fun(cand)`: strict <, in place
//   stochopy/optimize/_common.py:27-106       the population wrapper `fun(X)` itself is the caller's callable
#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"

using namespace sx;

namespace {

constexpr int kStep = 4;

// candidates of one generation -> cand (P, n) row-major; nothing else is touched
template <int RNG, int LPR>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave) void de_propose_kernel(const sx_de_args a,
                                                                              double *__restrict__ cand) {
    const sx_state *st = a.state;
    if (st->done) return;
    const int n = a.n;
    const int64_t P = a.P, ld = a.ld;
    const RowIds<LPR> id(P);
    if (!id.active) return;
    const int l = id.l;
    const int64_t row = id.row, it = st->it;
    const uint32_t gen = (uint32_t)(it + 1), grow = (uint32_t)(a.row0 + row);
    const double *__restrict__ cur = (it & 1) ? a.buf1 : a.buf0;
    const double *__restrict__ xi = cur + row * ld;
    const int strategy = a.strategy, k = donors_of(strategy);
    const bool repair = a.constraints != 0;
    int64_t d[kMaxDonors];
    int irand;
    if (RNG == SX_RNG_PHILOX) {
        philox_donors(P, k, row, grow, gen, a.key0, a.key1, n, d, irand);
    } else {
#pragma unroll
        for (int t = 0; t < kMaxDonors; ++t) d[t] = t < k ? (int64_t)a.donors[(int64_t)t * P + row] : 0;
        irand = a.irand[row];
    }
    const double *__restrict__ gb = a.gbest != nullptr ? a.gbest : cur + st->gbidx * ld;
    const double F = a.F, CR = a.CR;
    double *__restrict__ out = cand + row * (int64_t)n;
    const int nq = (n + LPR - 1) / LPR;
    for (int q0 = 0; q0 < nq; q0 += kStep) {  // four row steps per batch: one Philox call, loads in flight together
        double x[kStep], dv[kMaxDonors][kStep], g[kStep], r[kStep], rs[kStep];
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            const bool in = e < n;
            x[t] = in ? xi[e] : 0.0;
            g[t] = in ? gb[e] : 0.0;
#pragma unroll
            for (int s = 0; s < kMaxDonors; ++s) dv[s][t] = (s < k && in) ? cur[d[s] * ld + e] : 0.0;
            r[t] = (RNG == SX_RNG_HOST && in) ? a.r1[row * (int64_t)n + e] : 2.0;
            rs[t] = (RNG == SX_RNG_HOST && in && repair) ? a.resample[row * (int64_t)n + e] : 0.0;
        }
        if (RNG == SX_RNG_PHILOX) {
#pragma unroll
            for (int t = 0; t < kStep; t += 2) {  // 53-bit crossover uniforms, the fused kernel's layout
                const U4 w = philox4x32_10((uint32_t)((q0 + t) >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposeDeCross,
                                           a.key0, a.key1);
                r[t] = u53(w.x, w.y);
                r[t + 1] = u53(w.z, w.w);
            }
        }
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (e >= n) continue;
            const double v = de_mutant(strategy, g[t], dv[0][t], dv[1][t], dv[2][t], dv[3][t], dv[4][t], F);
            double c = (e == irand || r[t] <= CR) ? v : x[t];  // de/_de.py:341-344
            if (repair && (c < a.lower[e] || c > a.upper[e]))  // de/_constraints.py:21-26
                c = RNG == SX_RNG_HOST ? rs[t]
                                       : a.lower[e] + (a.upper[e] - a.lower[e]) *
                                             philox_u53(e, LPR, grow, gen, kPurposeDeResample, a.key0, a.key1);
            out[e] = c;
        }
    }
}

// V = w*V + c1*r1*(pbest - X) + c2*r2*(gbest - X); X += V (after Shrink): X and V in place
template <int RNG, int LPR>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave) void pso_move_kernel(const sx_pso_args a, const int has_stash) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const sx_state *st = a.state;
    if (st->done) return;
    const int n = a.n;
    const int64_t P = a.P, ld = a.ld;
    const RowIds<LPR> id(P);
    const int l = id.l;
    const int64_t rowc = id.rowc;
    double *Vn = lds + id.slot * (size_t)n;  // Shrink: the raw velocity waits here for the row-wide beta
    const uint32_t gen = (uint32_t)(st->it + 1), grow = (uint32_t)(a.row0 + rowc);
    double *xr = a.X + rowc * ld, *vr = a.V + rowc * ld;
    const double *pb = a.pbest + rowc * ld, *gb = a.gbest;
    const double w = a.w, c1 = a.c1, c2 = a.c2;
    const bool shrink = a.constraints != 0;
    double beta = __builtin_huge_val();
    const int nq = (n + LPR - 1) / LPR;
    // wide rows (n > sx_wide_from(): one wavefront per row, no dynamic LDS -- has_stash = 0): with Shrink the raw velocities are
    // formed AGAIN behind the row-wide beta instead of waiting in LDS -- same operands, same operations, same bits
    const bool stash = shrink && has_stash;
    // raw velocity of the two elements (q0 + t) * LPR + l, t = 0, 1 (one Philox call)
    auto raw = [&](int q0, double(&x)[2], double(&vn)[2]) {
        U4 pw = {0u, 0u, 0u, 0u}, pv = {0u, 0u, 0u, 0u};  // 53-bit r1 / r2, the fused kernel's layout
        if (RNG == SX_RNG_PHILOX) {
            pw = philox4x32_10((uint32_t)(q0 >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposePsoR1, a.key0, a.key1);
            pv = philox4x32_10((uint32_t)(q0 >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposePsoR2, a.key0, a.key1);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = (q0 + t) * LPR + l;
            x[t] = 0.0, vn[t] = 0.0;
            if (e >= n) continue;
            const double v = vr[e], p = pb[e], g = gb[e];
            x[t] = xr[e];
            double r1 = t ? u53(pw.z, pw.w) : u53(pw.x, pw.y), r2 = t ? u53(pv.z, pv.w) : u53(pv.x, pv.y);
            if (RNG == SX_RNG_HOST) {
                r1 = a.r1[rowc * (int64_t)n + e];
                r2 = a.r2[rowc * (int64_t)n + e];
            }
            vn[t] = pso_velocity(w, v, c1, r1, p, x[t], c2, r2, g);
        }
    };
    for (int q0 = 0; q0 < nq; q0 += 2) {  // two row steps share one Philox call
        double x[2], vn[2];
        raw(q0, x, vn);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (e >= n) continue;
            if (shrink) {  // cpso/_constraints.py:22-50
                if (stash) Vn[e] = vn[t];
                const double xc = x[t] + vn[t], lo = a.lower[e], hi = a.upper[e];
                if (xc < lo) beta = fmin(beta, (lo - x[t]) / vn[t]);
                if (xc > hi) beta = fmin(beta, (hi - x[t]) / vn[t]);
            } else if (id.active) {
                vr[e] = vn[t];
                xr[e] = x[t] + vn[t];
            }
        }
    }
    if (shrink) {
        beta = row_min<LPR>(beta);
        if (beta == __builtin_huge_val()) beta = 1.0;
        if (stash) {
            for (int e = l; e < n; e += LPR) {  // own elements only
                const double vn = Vn[e] * beta;
                if (id.active) {
                    const double xn = xr[e] + vn;
                    vr[e] = vn;
                    xr[e] = xn;
                }
            }
        } else {
            for (int q0 = 0; q0 < nq; q0 += 2) {
                double x[2], vn[2];
                raw(q0, x, vn);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int e = (q0 + t) * LPR + l;
                    if (e >= n || !id.active) continue;
                    const double vb = vn[t] * beta;
                    vr[e] = vb;
                    xr[e] = x[t] + vb;
                }
            }
        }
    }
}

// selection_sync after the evaluation (_common.py:127-129): rows with f < xfun take the candidate;
// xout may be xin (in place: PSO's pbest) or the other population buffer (DE); + the workgroup's best record
template <int LPR>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave) void rows_select_kernel(
    const double *__restrict__ cand, int64_t ldc, const double *__restrict__ f, const double *xin, double *xout,
    int64_t ldx, double *__restrict__ xfun, double *__restrict__ candfit, int64_t P, int n, const sx_state *st,
    double *__restrict__ part_f, int64_t *__restrict__ part_i) {
    __shared__ double sf[kMaxRowsPerBlock];
    __shared__ int64_t si[kMaxRowsPerBlock];
    if (st->done) return;
    const RowIds<LPR> id(P);
    const double fc = f[id.rowc], fold = xfun[id.rowc];
    const bool better = fc < fold;  // strict <
    if (id.active) {
        const double *src = better ? cand + id.row * ldc : xin + id.row * ldx;
        double *dst = xout + id.row * ldx;
        if (better || xin != xout)
            for (int e = id.l; e < n; e += LPR) dst[e] = src[e];
        if (id.l == 0) {
            if (better) xfun[id.row] = fc;
            if (candfit != nullptr) candfit[id.row] = fc;
        }
    }
    block_partial<LPR>(better ? fc : fold, id, sf, si, part_f, part_i);
}

}  // namespace

extern "C" int sx_de_propose(const sx_de_args *a, double *cand, void *stream) {
    SX_REQUIRE(a != nullptr && cand != nullptr, "sx_de_propose: null pointer");
    SX_REQUIRE(a->buf0 && a->buf1 && a->state, "sx_de_propose: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->P < (int64_t)1 << 31 && a->n >= 1 && a->ld >= a->n, "sx_de_propose: bad shape");
    SX_REQUIRE(a->strategy >= 0 && a->strategy <= SX_DE_BEST2BIN, "sx_de_propose: unknown strategy");
    SX_REQUIRE(a->P - 1 >= donors_of(a->strategy), "sx_de_propose: population too small for the strategy");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_de_propose: unknown rng mode");
    SX_REQUIRE(a->rng != SX_RNG_HOST || (a->r1 && a->donors && a->irand), "sx_de_propose: host draws missing");
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper && (a->rng != SX_RNG_HOST || a->resample)),
               "sx_de_propose: bounds / resample draws missing");
    const Geometry g = row_geometry(a->P, a->n);
    if (a->rng == SX_RNG_PHILOX) {
        SX_DISPATCH_LPR(a->n, hipLaunchKernelGGL((de_propose_kernel<SX_RNG_PHILOX, LPR>), dim3(g.blocks), dim3(g.threads), 0,
                                                 (hipStream_t)stream, *a, cand))
    } else {
        SX_DISPATCH_LPR(a->n, hipLaunchKernelGGL((de_propose_kernel<SX_RNG_HOST, LPR>), dim3(g.blocks), dim3(g.threads), 0,
                                                 (hipStream_t)stream, *a, cand))
    }
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_pso_move(const sx_pso_args *a, void *stream) {
    SX_REQUIRE(a != nullptr, "sx_pso_move: null args");
    SX_REQUIRE(a->X && a->V && a->pbest && a->gbest && a->state, "sx_pso_move: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->ld >= a->n, "sx_pso_move: bad shape");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_pso_move: unknown rng mode");
    SX_REQUIRE(a->rng != SX_RNG_HOST || (a->r1 && a->r2), "sx_pso_move: host draws missing");
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper), "sx_pso_move: bounds missing");
    const Geometry g = row_geometry(a->P, a->n);
    const size_t lds = a->n > sx::wide_from() ? 0 : (size_t)rows_per_block(a->n) * a->n * sizeof(double);
    if (a->rng == SX_RNG_PHILOX) {
        SX_DISPATCH_LPR(a->n, hipLaunchKernelGGL((pso_move_kernel<SX_RNG_PHILOX, LPR>), dim3(g.blocks), dim3(g.threads), lds,
                                                 (hipStream_t)stream, *a, lds != 0 ? 1 : 0))
    } else {
        SX_DISPATCH_LPR(a->n, hipLaunchKernelGGL((pso_move_kernel<SX_RNG_HOST, LPR>), dim3(g.blocks), dim3(g.threads), lds,
                                                 (hipStream_t)stream, *a, lds != 0 ? 1 : 0))
    }
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_rows_select(const double *cand, int64_t ldc, const double *f, const double *xin, double *xout,
                              int64_t ldx, double *xfun, double *candfit, int64_t P, int n, const sx_state *state,
                              double *part_f, int64_t *part_i, void *stream) {
    SX_REQUIRE(cand && f && xin && xout && xfun && state && part_f && part_i, "sx_rows_select: null pointer");
    SX_REQUIRE(P >= 1 && n >= 1 && ldc >= n && ldx >= n, "sx_rows_select: bad shape");
    const Geometry g = row_geometry(P, n);
    SX_DISPATCH_LPR(n, hipLaunchKernelGGL((rows_select_kernel<LPR>), dim3(g.blocks), dim3(g.threads), 0, (hipStream_t)stream,
                                          cand, ldc, f, xin, xout, ldx, xfun, candfit, P, n, state, part_f, part_i))
    SX_LAUNCH_CHECK();
    return 0;
}
