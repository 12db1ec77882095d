// updating="immediate": the reference's asynchronous generations, where individual i already sees what
// individuals 0..i-1 did in the SAME generation (their accepted rows as donors, their improvements of the
// global best).  That order is the algorithm, so a generation is one sequential sweep: ONE row group
// (LPR lanes, the same element <-> lane layout and Philox counters as the synchronous kernels) walks the
// population; the trial row is staged in LDS, the objective uses the same numpy-order reduction, the best
// row lives in LDS.  Every global element is only ever touched by the lane that owns it, so plain loads and
// stores are ordered by the program order of that lane.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/de/_de.py:354-391       de_async (mutation, crossover, constraint per individual)
//   stochopy/optimize/cpso/_cpso.py:364-402   pso_async
//   stochopy/optimize/_common.py:163-194      selection_async (<=, best/status update per individual;
//                                             only the LAST individual's status survives the sweep)
//   stochopy/optimize/cpso/_constraints.py:56-64  Shrink, one-row form
#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"

namespace sx {
int make_plan_arg(int fun_id, int n, PlanArg *out);
}
using namespace sx;

namespace {

// 32-bit uniform of element e of a row: the value the synchronous kernels draw for it
// (slot = (q >> 2) * LPR + l, word = q & 3; de kernel) / (slot = (q >> 1) * LPR + l, words by parity; pso kernel)
template <int LPR>
__device__ __forceinline__ double de_cross_uniform(int e, uint32_t grow, uint32_t gen, uint32_t k0, uint32_t k1) {
    const uint32_t q = (uint32_t)e / (uint32_t)LPR, l = (uint32_t)e & (uint32_t)(LPR - 1);
    const U4 w = philox4x32_10((q >> 2) * (uint32_t)LPR + l, grow, gen, kPurposeDeCross, k0, k1);
    const uint32_t s = q & 3u;
    return u32(s == 0 ? w.x : s == 1 ? w.y : s == 2 ? w.z : w.w);
}
template <int LPR>
__device__ __forceinline__ void pso_uniforms(int e, uint32_t grow, uint32_t gen, uint32_t k0, uint32_t k1, double &r1,
                                             double &r2) {
    const uint32_t q = (uint32_t)e / (uint32_t)LPR, l = (uint32_t)e & (uint32_t)(LPR - 1);
    const U4 w = philox4x32_10((q >> 1) * (uint32_t)LPR + l, grow, gen, kPurposePsoR1, k0, k1);
    r1 = u32((q & 1u) ? w.z : w.x);
    r2 = u32((q & 1u) ? w.w : w.y);
}

__host__ __device__ inline bool best_in_lds(int n) { return (size_t)(lds_row_stride(n) + n) * sizeof(double) <= 64 * 1024; }

// selection_async's best update (_common.py:174-192): returns the status this individual leaves behind
template <int LPR>
__device__ __forceinline__ int improve_best(const double *U, double *G, int n, int l, double fc, double &gfit,
                                            double xtol, double ftol) {
    double acc = 0.0;
    for (int e = l; e < n; e += LPR) {
        const double d = G[e] - U[e];
        acc += d * d;
        G[e] = U[e];
    }
    const double dx = sqrt(row_sum<LPR>(acc));
    gfit = fc;
    if (fc <= ftol) return dx <= xtol ? 0 : 1;
    return SX_STATUS_NONE;
}

__device__ __forceinline__ void publish(sx_state *st, int64_t it, double gfit, int status, int maxiter) {
    if (status == SX_STATUS_NONE && it >= maxiter) status = -1;  // de/_de.py:387-388
    st->it = it;
    st->gfit = gfit;
    st->dx = 0.0;
    st->status = status;
    st->done = status != SX_STATUS_NONE;
}

template <int FUN, int RNG, int LPR>
__global__ __launch_bounds__(kWave) void de_async_kernel(const sx_de_args a, const PlanArg plan) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sx_state *st = a.state;
    if (st->done) return;
    const int n = a.n, l = (int)threadIdx.x;  // blockDim.x == LPR: one row group
    const int64_t P = a.P, ld = a.ld;
    // the best row: in LDS behind the staging area while both fit in 64 KiB, else in place in a.gbest
    double *U = lds, *G = best_in_lds(n) ? lds + lds_row_stride(n) : a.gbest;
    const int64_t it = st->it + 1;  // the generation this sweep produces
    const uint32_t gen = (uint32_t)it;
    double gfit = st->gfit;
    if (G != a.gbest)
        for (int e = l; e < n; e += LPR) G[e] = a.gbest[e];
    const int strategy = a.strategy, k = donors_of(strategy);
    const bool repair = a.constraints != 0;
    const double F = a.F, CR = a.CR;
    double *X = a.buf0;  // ONE population, updated in place as the sweep goes
    int status = SX_STATUS_NONE;
    lds_wave_fence();

    for (int64_t i = 0; i < P; ++i) {
        const uint32_t grow = (uint32_t)(a.row0 + i);
        int64_t d[kMaxDonors];
        int irand;
        if (RNG == SX_RNG_PHILOX) {
            philox_donors(P, k, i, grow, gen, a.key0, a.key1, n, d, irand);
        } else {
#pragma unroll
            for (int t = 0; t < kMaxDonors; ++t) d[t] = t < k ? (int64_t)a.donors[(int64_t)t * P + i] : 0;
            irand = a.irand[i];
        }
        const double *xi = X + i * ld;
        const double fold = a.fit[i];
#pragma unroll 4
        for (int e = l; e < n; e += LPR) {
            double dv[kMaxDonors];
#pragma unroll
            for (int t = 0; t < kMaxDonors; ++t) dv[t] = t < k ? X[d[t] * ld + e] : 0.0;
            const double x = xi[e], g = G[e];
            double v;  // de/_strategy.py, same association
            if (strategy == SX_DE_BEST1BIN)
                v = g + F * (dv[0] - dv[1]);
            else if (strategy == SX_DE_RAND1BIN)
                v = dv[0] + F * (dv[1] - dv[2]);
            else if (strategy == SX_DE_BEST2BIN)
                v = g + F * (((dv[0] + dv[1]) - dv[2]) - dv[3]);
            else
                v = dv[0] + F * (((dv[1] + dv[2]) - dv[3]) - dv[4]);
            const double r = RNG == SX_RNG_HOST ? a.r1[i * (int64_t)n + e] : de_cross_uniform<LPR>(e, grow, gen, a.key0, a.key1);
            double cand = (e == irand || r <= CR) ? v : x;  // de/_de.py:381-384
            if (repair && (cand < a.lower[e] || cand > a.upper[e]))
                cand = RNG == SX_RNG_HOST ? a.resample[i * (int64_t)n + e]
                                          : a.lower[e] + (a.upper[e] - a.lower[e]) *
                                                philox_u53(e, LPR, grow, gen, kPurposeDeResample, a.key0, a.key1);
            U[e] = cand;
        }
        const double fc = row_objective<FUN, LPR>(U, n, plan, l);
        status = SX_STATUS_NONE;  // each individual overwrites the status (_common.py:168)
        if (fc <= fold) {         // _common.py:169: <=, unlike the synchronous selection
            double *xo = X + i * ld;
            for (int e = l; e < n; e += LPR) xo[e] = U[e];
            if (l == 0) a.fit[i] = fc;
            if (fc <= gfit) status = improve_best<LPR>(U, G, n, l, fc, gfit, a.xtol, a.ftol);
        }
        if (l == 0 && a.candfit != nullptr) a.candfit[i] = fc;
        lds_wave_fence();  // U and G settled before the next individual rewrites / reads them
    }
    if (G != a.gbest)
        for (int e = l; e < n; e += LPR) a.gbest[e] = G[e];
    if (l == 0) publish(st, it, gfit, status, a.maxiter);
}

template <int FUN, int RNG, int LPR>
__global__ __launch_bounds__(kWave) void pso_async_kernel(const sx_pso_args a, const PlanArg plan) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sx_state *st = a.state;
    if (st->done) return;
    const int n = a.n, l = (int)threadIdx.x;
    const int64_t P = a.P, ld = a.ld;
    double *U = lds, *G = best_in_lds(n) ? lds + lds_row_stride(n) : a.gbest;
    const int64_t it = st->it + 1;
    const uint32_t gen = (uint32_t)it;
    double gfit = st->gfit;
    if (G != a.gbest)
        for (int e = l; e < n; e += LPR) G[e] = a.gbest[e];
    const bool shrink = a.constraints != 0;
    const double w = a.w, c1 = a.c1, c2 = a.c2;
    int status = SX_STATUS_NONE;
    lds_wave_fence();

    for (int64_t i = 0; i < P; ++i) {
        const uint32_t grow = (uint32_t)(a.row0 + i);
        double *xr = a.X + i * ld, *vr = a.V + i * ld, *pb = a.pbest + i * ld;
        const double fold = a.pbestfit[i];
        double beta = __builtin_huge_val();
#pragma unroll 4
        for (int e = l; e < n; e += LPR) {
            const double x = xr[e], v = vr[e], p = pb[e], g = G[e];
            double r1, r2;
            if (RNG == SX_RNG_HOST) {
                r1 = a.r1[i * (int64_t)n + e];
                r2 = a.r2[i * (int64_t)n + e];
            } else {
                pso_uniforms<LPR>(e, grow, gen, a.key0, a.key1, r1, r2);
            }
            const double vn = (w * v + (c1 * r1) * (p - x)) + (c2 * r2) * (g - x);  // cpso/_cpso.py:326
            if (shrink) {  // cpso/_constraints.py:22-42, 56-64
                U[e] = vn;
                const double xc = x + vn, lo = a.lower[e], hi = a.upper[e];
                if (xc < lo) beta = fmin(beta, (lo - x) / vn);
                if (xc > hi) beta = fmin(beta, (hi - x) / vn);
            } else {
                const double xn = x + vn;
                U[e] = xn;
                vr[e] = vn;
                xr[e] = xn;
            }
        }
        if (shrink) {
            beta = row_min<LPR>(beta);
            if (beta == __builtin_huge_val()) beta = 1.0;
            for (int e = l; e < n; e += LPR) {  // own elements only: no fence needed
                const double vn = U[e] * beta;
                const double xn = xr[e] + vn;
                U[e] = xn;
                vr[e] = vn;
                xr[e] = xn;
            }
        }
        const double fc = row_objective<FUN, LPR>(U, n, plan, l);
        status = SX_STATUS_NONE;
        if (fc <= fold) {
            for (int e = l; e < n; e += LPR) pb[e] = U[e];
            if (l == 0) a.pbestfit[i] = fc;
            if (fc <= gfit) status = improve_best<LPR>(U, G, n, l, fc, gfit, a.xtol, a.ftol);
        }
        if (l == 0 && a.candfit != nullptr) a.candfit[i] = fc;
        lds_wave_fence();
    }
    if (G != a.gbest)
        for (int e = l; e < n; e += LPR) a.gbest[e] = G[e];
    if (l == 0) publish(st, it, gfit, status, a.maxiter);
}

typedef void (*de_async_t)(const sx_de_args, const PlanArg);
typedef void (*pso_async_t)(const sx_pso_args, const PlanArg);

template <int RNG, int LPR>
de_async_t pick_de(int fun_id) {
    switch (fun_id) {
        case SX_FUN_ACKLEY: return de_async_kernel<SX_FUN_ACKLEY, RNG, LPR>;
        case SX_FUN_GRIEWANK: return de_async_kernel<SX_FUN_GRIEWANK, RNG, LPR>;
        case SX_FUN_QUARTIC: return de_async_kernel<SX_FUN_QUARTIC, RNG, LPR>;
        case SX_FUN_RASTRIGIN: return de_async_kernel<SX_FUN_RASTRIGIN, RNG, LPR>;
        case SX_FUN_ROSENBROCK: return de_async_kernel<SX_FUN_ROSENBROCK, RNG, LPR>;
        case SX_FUN_SPHERE: return de_async_kernel<SX_FUN_SPHERE, RNG, LPR>;
        case SX_FUN_STYBLINSKI_TANG: return de_async_kernel<SX_FUN_STYBLINSKI_TANG, RNG, LPR>;
    }
    return nullptr;
}
template <int RNG, int LPR>
pso_async_t pick_pso(int fun_id) {
    switch (fun_id) {
        case SX_FUN_ACKLEY: return pso_async_kernel<SX_FUN_ACKLEY, RNG, LPR>;
        case SX_FUN_GRIEWANK: return pso_async_kernel<SX_FUN_GRIEWANK, RNG, LPR>;
        case SX_FUN_QUARTIC: return pso_async_kernel<SX_FUN_QUARTIC, RNG, LPR>;
        case SX_FUN_RASTRIGIN: return pso_async_kernel<SX_FUN_RASTRIGIN, RNG, LPR>;
        case SX_FUN_ROSENBROCK: return pso_async_kernel<SX_FUN_ROSENBROCK, RNG, LPR>;
        case SX_FUN_SPHERE: return pso_async_kernel<SX_FUN_SPHERE, RNG, LPR>;
        case SX_FUN_STYBLINSKI_TANG: return pso_async_kernel<SX_FUN_STYBLINSKI_TANG, RNG, LPR>;
    }
    return nullptr;
}

size_t async_lds(int n) { return (size_t)(lds_row_stride(n) + (best_in_lds(n) ? n : 0)) * sizeof(double); }

}  // namespace

// One asynchronous DE generation (the whole sweep + status), population a->buf0 in place, best row a->gbest
// (in/out), a->state: it, gfit in/out; status, done out.  a->buf1 / part_f / part_i are not used.
extern "C" int sx_de_async_generation(const sx_de_args *a, void *stream) {
    SX_REQUIRE(a != nullptr, "sx_de_async: null args");
    SX_REQUIRE(a->buf0 && a->fit && a->gbest && a->state, "sx_de_async: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->P < (int64_t)1 << 31 && a->n >= 1 && a->n <= kMaxDim && a->ld >= a->n, "sx_de_async: bad shape");
    SX_REQUIRE(a->fun_id >= 0 && a->fun_id < SX_FUN_COUNT, "sx_de_async: unknown objective");
    SX_REQUIRE(a->strategy >= 0 && a->strategy <= SX_DE_BEST2BIN, "sx_de_async: unknown strategy");
    SX_REQUIRE(a->P - 1 >= donors_of(a->strategy), "sx_de_async: population too small for the strategy");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_de_async: unknown rng mode");
    SX_REQUIRE(a->rng != SX_RNG_HOST || (a->r1 && a->donors && a->irand), "sx_de_async: host draws missing");
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper && (a->rng != SX_RNG_HOST || a->resample)),
               "sx_de_async: bounds / resample draws missing");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    de_async_t kern = nullptr;
    if (a->rng == SX_RNG_PHILOX) {
        SX_DISPATCH_LPR(a->n, kern = (pick_de<SX_RNG_PHILOX, LPR>(a->fun_id)))
    } else {
        SX_DISPATCH_LPR(a->n, kern = (pick_de<SX_RNG_HOST, LPR>(a->fun_id)))
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(lanes_per_row(a->n)), async_lds(a->n), (hipStream_t)stream, *a, plan);
    SX_LAUNCH_CHECK();
    return 0;
}

// One asynchronous PSO generation: X, V, pbest, pbestfit in place, a->gbest in/out, a->state as above.
extern "C" int sx_pso_async_generation(const sx_pso_args *a, void *stream) {
    SX_REQUIRE(a != nullptr, "sx_pso_async: null args");
    SX_REQUIRE(a->X && a->V && a->pbest && a->pbestfit && a->gbest && a->state, "sx_pso_async: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->n <= kMaxDim && a->ld >= a->n, "sx_pso_async: bad shape");
    SX_REQUIRE(a->fun_id >= 0 && a->fun_id < SX_FUN_COUNT, "sx_pso_async: unknown objective");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_pso_async: unknown rng mode");
    SX_REQUIRE(a->rng != SX_RNG_HOST || (a->r1 && a->r2), "sx_pso_async: host draws missing");
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper), "sx_pso_async: bounds missing");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    pso_async_t kern = nullptr;
    if (a->rng == SX_RNG_PHILOX) {
        SX_DISPATCH_LPR(a->n, kern = (pick_pso<SX_RNG_PHILOX, LPR>(a->fun_id)))
    } else {
        SX_DISPATCH_LPR(a->n, kern = (pick_pso<SX_RNG_HOST, LPR>(a->fun_id)))
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(lanes_per_row(a->n)), async_lds(a->n), (hipStream_t)stream, *a, plan);
    SX_LAUNCH_CHECK();
    return 0;
}
