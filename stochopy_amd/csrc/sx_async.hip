// updating="immediate": the reference's asynchronous generations, where individual i already sees what
// individuals 0..i-1 did in the SAME generation (their accepted rows as donors, their improvements of the
// global best).  That order is the algorithm, so a generation is one ordered sweep by ONE workgroup (same
// element <-> lane layout, Philox counters and numpy-order objective as the synchronous kernels; trial rows and
// the best row live in LDS).  The sweep is still parallel inside: rounds of up to 64 individuals are proposed
// together and then judged in order, and only the few that really depended on an earlier one of their round
// are recomputed (see below) -- the outcome is the sequential one, bit for bit.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/de/_de.py:354-391       de_async (mutation, crossover, constraint per individual)
//   stochopy/optimize/cpso/_cpso.py:364-402   pso_async
//   stochopy/optimize/_common.py:163-194      selection_async (<=, best/status update per individual;
//                                             only the LAST individual's status survives the sweep)
//   stochopy/optimize/cpso/_constraints.py:56-64  Shrink, one-row form
// (the ordered sweeps sit at their 128-VGPR cap: the reduction keeps the form with fewer live values -- tools/isa_survey.py)
// Two translation units (the longest compile of the library otherwise, ~95 s, the build's critical path): this file compiles the DE
// sweeps; sx_async_pso.hip includes it with SX_ASYNC_PART = 1 for the PSO / CPSO sweeps.
#ifndef SX_ASYNC_PART
#define SX_ASYNC_PART 0
#endif
#define SX_FUSED_TAIL_BY_LANE 0
#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"

namespace sx {
int make_plan_arg(int fun_id, int n, PlanArg *out);
}
using namespace sx;

namespace {

// ---------------------------------------------------------------------------
// Exact sweeps, B individuals at a time.  ONE workgroup; every row group (LPR lanes) owns a slot.  A round
//   1. computes the trial rows and fitness values of individuals i0..i0+B-1 in parallel, as if none of them
//      influenced another (on the population / best row as they stand after the previous round);
//   2. judges them in order, lane-parallel (lane j of every wave holds individual j; every wave derives the same
//      decisions): the accept mask is one ballot; individual j is "dirty" if one of its donors is accepted
//      earlier in this round (dep & accepted & below(j)), or (strategies / PSO that read the best row) the best
//      row improves earlier in this round.  The longest run that is valid on the values at hand is committed by
//      selection_async's rules; if it ends before the round does, the rest of the round is proposed again,
//      together, on the state as it stands then.
// The result is the sequential sweep's, bit for bit; conflicts are rare (k*B/(2P) donors per individual, a
// handful of best-row improvements per generation), and each costs one more parallel proposal.
// ---------------------------------------------------------------------------
#ifndef SX_SWEEP_WAVES
#define SX_SWEEP_WAVES 16  // (a build-time knob for A/B measurements)
#endif
constexpr int kSweepWaves = SX_SWEEP_WAVES;
// Ackley's and Griewank's term code (fp64 cos, exp) does not fit the 128 VGPRs a 16-wave workgroup leaves a lane: those
// kernels spilled 220-350 bytes per lane to scratch; with 8 waves (256 VGPRs) they do not, and half the slots per round
// still make the faster sweep (DE Ackley n=128 P=4096 1.89 -> 1.52 ms per generation, PSO Ackley n=256 2.90 -> 2.54;
// Rosenbrock, which spills 48 bytes, is 20 % slower with 8 waves and keeps 16)
__host__ __device__ constexpr int sweep_waves(int fun_id) {
    return (fun_id == SX_FUN_ACKLEY || fun_id == SX_FUN_GRIEWANK) ? (kSweepWaves + 1) / 2 : kSweepWaves;
}
constexpr int kSweepSlots = 64;  // <= 64: the "replaced in this round" set is one 64-bit mask

struct SweepGeometry {
    int waves, slots;
    size_t lds;
};
// slot = staging row (+ `extra` more rows of n doubles); the best row G behind the slots
inline SweepGeometry sweep_geometry(int n, int extra, int fun_id) {
    const int rpw = kWave / lanes_per_row(n);
    const size_t slot_bytes = (size_t)(lds_row_stride(n) + extra * n) * sizeof(double);
    const size_t avail = 160 * 1024 - 4096 - (size_t)n * sizeof(double);  // 160 KB per CU, static arrays, G
    int waves = (int)(avail / slot_bytes) / rpw;
    waves = waves < 1 ? 1 : (waves > sweep_waves(fun_id) ? sweep_waves(fun_id) : waves);
    return SweepGeometry{waves, waves * rpw, (size_t)waves * rpw * slot_bytes + (size_t)n * sizeof(double)};
}

// value of lane j (uniform) of a 64-bit quantity: two v_readlane, no LDS round trip
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int j) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int j) {
    return __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), j));
}

// selection_async's best update (_common.py:174-192): returns the status this individual leaves behind
template <int LPR>
__device__ __forceinline__ int improve_best(const double *U, double *G, int n, int l, double fc, double xtol,
                                            double ftol) {
    double acc = 0.0;
    for (int e = l; e < n; e += LPR) {
        const double d = G[e] - U[e];
        acc += d * d;
        G[e] = U[e];
    }
    const double dx = sqrt(row_sum<LPR>(acc));
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

// NFIX: the row length when it is exactly 4 * LPR (64, 128, 256) and the draws are made in the kernel -- a compile-time
// constant, and with it numpy's summation plan (sx_device.hpp row_reduce_fixed / row_reduce_static); 0 otherwise.
template <int FUN, int RNG, int LPR, int NFIX = 0>
__global__ __launch_bounds__(sweep_waves(FUN) * kWave) void de_async_kernel(const sx_de_args a, const PlanArg plan) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sfc[kSweepSlots], sfold[kSweepSlots];
    __shared__ unsigned long long sdep[kSweepSlots];
    __shared__ int sstatus;
    sx_state *st = a.state;
    if (st->done) return;
    constexpr int RPW = kWave / LPR;
    const int n = NFIX ? NFIX : a.n, lane = (int)(threadIdx.x & 63), l = lane & (LPR - 1);
    const int slot = (int)(threadIdx.x >> 6) * RPW + lane / LPR, B = (int)(blockDim.x >> 6) * RPW;
    const int64_t P = a.P, ld = a.ld;
    const int stride = lds_row_stride(n);
    double *U = lds + (size_t)slot * stride, *G = lds + (size_t)B * stride;
    const int64_t it = st->it + 1;  // the generation this sweep produces
    const uint32_t gen = (uint32_t)it;
    double gfit = st->gfit;  // uniform: every thread applies the same updates
    for (int e = (int)threadIdx.x; e < n; e += (int)blockDim.x) G[e] = a.gbest[e];
    const int strategy = a.strategy, k = donors_of(strategy);
    const bool repair = a.constraints != 0;
    const bool use_best = strategy == SX_DE_BEST1BIN || strategy == SX_DE_BEST2BIN;
    const double F = a.F, CR = a.CR;
    double *X = a.buf0;  // ONE population, updated in place as the sweep goes
    int status = SX_STATUS_NONE;
    __syncthreads();

    // trial row of individual i into this group's slot + its fitness (de/_de.py:376-384); `dep` = the slots of
    // this round that hold one of its donors
    auto propose = [&](int64_t i, int64_t i0, unsigned long long &dep) -> double {
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
        dep = 0ull;
#pragma unroll
        for (int t = 0; t < kMaxDonors; ++t)
            if (t < k && d[t] >= i0 && d[t] < i) dep |= 1ull << (int)(d[t] - i0);
        const double *xi = X + i * ld;
        const int nq = (n + LPR - 1) / LPR;
        for (int q0 = 0; q0 < nq; q0 += 4) {  // four row steps share one Philox call, as in the synchronous kernel
            double rr[4] = {2.0, 2.0, 2.0, 2.0};
            if (RNG == SX_RNG_PHILOX) {
#pragma unroll
                for (int t = 0; t < 4; t += 2) {  // 53-bit crossover uniforms, the synchronous kernel's layout
                    const U4 w = philox4x32_10((uint32_t)((q0 + t) >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen,
                                               kPurposeDeCross, a.key0, a.key1);
                    rr[t] = u53(w.x, w.y);
                    rr[t + 1] = u53(w.z, w.w);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (e >= n) continue;
            double dv[kMaxDonors];
#pragma unroll
            for (int t = 0; t < kMaxDonors; ++t) dv[t] = t < k ? X[d[t] * ld + e] : 0.0;
            const double x = xi[e], g = G[e];
            const double v = de_mutant(strategy, g, dv[0], dv[1], dv[2], dv[3], dv[4], F);
            const double r = RNG == SX_RNG_HOST ? a.r1[i * (int64_t)n + e] : rr[t];
            double cand = (e == irand || r <= CR) ? v : x;  // de/_de.py:381-384
            if (repair && (cand < a.lower[e] || cand > a.upper[e]))
                cand = RNG == SX_RNG_HOST ? a.resample[i * (int64_t)n + e]
                                          : a.lower[e] + (a.upper[e] - a.lower[e]) *
                                                philox_u53(e, LPR, grow, gen, kPurposeDeResample, a.key0, a.key1);
            U[e] = cand;
            }
        }
        return row_objective<FUN, LPR, false, NFIX, 0>(U, n, plan, l);
    };

    for (int64_t i0 = 0; i0 < P; i0 += B) {
        const int nb = (int)(P - i0 < B ? P - i0 : B);
        if (slot < nb) {
            unsigned long long dep;
            const double fc = propose(i0 + slot, i0, dep);
            if (l == 0) {
                sfc[slot] = fc;
                sfold[slot] = a.fit[i0 + slot];
                sdep[slot] = dep;
            }
        }
        __syncthreads();
        // the walk, lane-parallel: lane j of every wave holds individual j of the round and every wave derives
        // the same decisions.  [start, stop) is the longest run that can be judged on the values at hand: it ends
        // before the first individual one of whose donors is accepted earlier in the run, and (strategies that
        // read the best row) after the first individual that improves the best.
        const double wfold = lane < nb ? sfold[lane] : 0.0;
        const unsigned long long wdep = lane < nb ? sdep[lane] : 0ull;
        const unsigned long long below = (1ull << lane) - 1ull;
        int start = 0;
        while (start < nb) {
            const double wfc = lane < nb ? sfc[lane] : 0.0;
            const bool in = lane >= start && lane < nb;
            const bool acc = in && wfc <= wfold;  // _common.py:169: <=, unlike the synchronous selection
            const unsigned long long accmask = __ballot(acc);
            const unsigned long long dirty = __ballot(in && (wdep & accmask & below) != 0ull);
            int stop = dirty ? (int)__ffsll((long long)dirty) - 1 : nb;
            // improvements of the best row inside the run, in order (_common.py:174-192)
            unsigned long long todo = __ballot(acc && wfc <= gfit) & (stop >= 64 ? ~0ull : (1ull << stop) - 1ull);
            int last_imp = -1, last_status = SX_STATUS_NONE;
            while (todo) {
                const int j = (int)__ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const double fj = readlane_f64(wfc, j);
                if (!(fj <= gfit)) continue;  // an earlier one of the run went lower
                __syncthreads();              // nobody still reads the best row
                if (slot == j) {
                    const int s = improve_best<LPR>(U, G, n, l, fj, a.xtol, a.ftol);
                    if (l == 0) sstatus = s;
                }
                __syncthreads();
                last_status = sstatus;
                last_imp = j;
                gfit = fj;
                if (use_best) {  // the individuals after j were proposed on the old best row
                    stop = j + 1;
                    break;
                }
            }
            // the run is final: accepted rows replace their parents
            if (slot >= start && slot < stop) {
                const int64_t i = i0 + slot;
                const double fc = sfc[slot];
                if ((accmask >> slot) & 1ull) {
                    double *xo = X + i * ld;
                    for (int e = l; e < n; e += LPR) xo[e] = U[e];
                    if (l == 0) a.fit[i] = fc;
                }
                if (l == 0 && a.candfit != nullptr) a.candfit[i] = fc;
            }
            status = last_imp == stop - 1 ? last_status : SX_STATUS_NONE;  // the status the LAST individual leaves
            start = stop;
            __syncthreads();  // rows and best row in place; slots [0, start) are done with
            if (start < nb) {  // propose the rest of the round again, together, on the state as it stands now
                if (slot >= start && slot < nb) {
                    unsigned long long dep;
                    const double fc = propose(i0 + slot, i0, dep);
                    if (l == 0) sfc[slot] = fc;
                }
                __syncthreads();
            }
        }
    }
    for (int e = (int)threadIdx.x; e < n; e += (int)blockDim.x) a.gbest[e] = G[e];
    if (threadIdx.x == 0) publish(st, it, gfit, status, a.maxiter);
}

template <int FUN, int RNG, int LPR, int NFIX = 0>
__global__ __launch_bounds__(sweep_waves(FUN) * kWave) void pso_async_kernel(const sx_pso_args a, const PlanArg plan) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sfc[kSweepSlots], sfold[kSweepSlots];
    __shared__ int sstatus;
    sx_state *st = a.state;
    if (st->done) return;
    constexpr int RPW = kWave / LPR;
    const int n = NFIX ? NFIX : a.n, lane = (int)(threadIdx.x & 63), l = lane & (LPR - 1);
    const int slot = (int)(threadIdx.x >> 6) * RPW + lane / LPR, B = (int)(blockDim.x >> 6) * RPW;
    const int64_t P = a.P, ld = a.ld;
    const int stride = lds_row_stride(n) + n;  // staging row + the new velocity
    double *U = lds + (size_t)slot * stride, *Vn = U + lds_row_stride(n), *G = lds + (size_t)B * stride;
    const int64_t it = st->it + 1;
    const uint32_t gen = (uint32_t)it;
    double gfit = st->gfit;
    for (int e = (int)threadIdx.x; e < n; e += (int)blockDim.x) G[e] = a.gbest[e];
    const bool shrink = a.constraints != 0;
    const double w = a.w, c1 = a.c1, c2 = a.c2;
    int status = SX_STATUS_NONE;
    __syncthreads();

    // new position (U) and velocity (Vn) of particle i against the best row as it stands + the fitness
    // (cpso/_cpso.py:324-329, 385-392); X, V, pbest are only written once the particle has been judged
    auto propose = [&](int64_t i) -> double {
        const uint32_t grow = (uint32_t)(a.row0 + i);
        const double *xr = a.X + i * ld, *vr = a.V + i * ld, *pb = a.pbest + i * ld;
        double beta = __builtin_huge_val();
        const int nq = (n + LPR - 1) / LPR;
        for (int q0 = 0; q0 < nq; q0 += 2) {  // two row steps share one Philox call, as in the synchronous kernel
            U4 pw = {0u, 0u, 0u, 0u}, pv = {0u, 0u, 0u, 0u};  // 53-bit r1 / r2, the synchronous kernel's layout
            if (RNG == SX_RNG_PHILOX) {
                pw = philox4x32_10((uint32_t)(q0 >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposePsoR1, a.key0, a.key1);
                pv = philox4x32_10((uint32_t)(q0 >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposePsoR2, a.key0, a.key1);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (e >= n) continue;
            const double x = xr[e], v = vr[e], p = pb[e], g = G[e];
            double r1 = t ? u53(pw.z, pw.w) : u53(pw.x, pw.y), r2 = t ? u53(pv.z, pv.w) : u53(pv.x, pv.y);
            if (RNG == SX_RNG_HOST) {
                r1 = a.r1[i * (int64_t)n + e];
                r2 = a.r2[i * (int64_t)n + e];
            }
            const double vn = pso_velocity(w, v, c1, r1, p, x, c2, r2, g);
            Vn[e] = vn;
            const double xc = x + vn;
            U[e] = xc;
            if (shrink) {  // cpso/_constraints.py:22-42, 56-64
                const double lo = a.lower[e], hi = a.upper[e];
                if (xc < lo) beta = fmin(beta, (lo - x) / vn);
                if (xc > hi) beta = fmin(beta, (hi - x) / vn);
            }
            }
        }
        if (shrink) {
            beta = row_min<LPR>(beta);
            if (beta == __builtin_huge_val()) beta = 1.0;
            for (int e = l; e < n; e += LPR) {  // own elements only: no fence needed
                const double vn = Vn[e] * beta;
                Vn[e] = vn;
                U[e] = xr[e] + vn;
            }
        }
        return row_objective<FUN, LPR, false, NFIX, 0>(U, n, plan, l);
    };

    for (int64_t i0 = 0; i0 < P; i0 += B) {
        const int nb = (int)(P - i0 < B ? P - i0 : B);
        if (slot < nb) {
            const double fc = propose(i0 + slot);
            if (l == 0) {
                sfc[slot] = fc;
                sfold[slot] = a.pbestfit[i0 + slot];
            }
        }
        __syncthreads();
        // the walk, lane-parallel (see de_async_kernel): particles only couple through the best row, so a run
        // [start, stop) ends right after the first particle that improves it
        const double wfold = lane < nb ? sfold[lane] : 0.0;
        int start = 0;
        while (start < nb) {
            const double wfc = lane < nb ? sfc[lane] : 0.0;
            const bool in = lane >= start && lane < nb;
            const bool acc = in && wfc <= wfold;
            const unsigned long long accmask = __ballot(acc);
            const unsigned long long imp = __ballot(acc && wfc <= gfit);
            int stop = nb, last_status = SX_STATUS_NONE;
            if (imp) {
                const int j = (int)__ffsll((long long)imp) - 1;
                const double fj = readlane_f64(wfc, j);
                __syncthreads();  // nobody still reads the best row
                if (slot == j) {
                    const int s = improve_best<LPR>(U, G, n, l, fj, a.xtol, a.ftol);
                    if (l == 0) sstatus = s;
                }
                __syncthreads();
                last_status = sstatus;
                gfit = fj;
                stop = j + 1;
            }
            // every particle of the run moves (cpso/_cpso.py:389); the accepted ones also become their own best
            if (slot >= start && slot < stop) {
                const int64_t i = i0 + slot;
                const bool mine = (accmask >> slot) & 1ull;
                double *xr = a.X + i * ld, *vr = a.V + i * ld, *pb = a.pbest + i * ld;
                for (int e = l; e < n; e += LPR) {
                    const double xn = U[e];
                    xr[e] = xn;
                    vr[e] = Vn[e];
                    if (mine) pb[e] = xn;
                }
                if (l == 0) {
                    if (mine) a.pbestfit[i] = sfc[slot];
                    if (a.candfit != nullptr) a.candfit[i] = sfc[slot];
                }
            }
            status = imp ? last_status : SX_STATUS_NONE;  // imp: the run's last particle is the improver
            start = stop;
            __syncthreads();
            if (start < nb) {  // the rest of the round against the new best row
                if (slot >= start && slot < nb) {
                    const double fc = propose(i0 + slot);
                    if (l == 0) sfc[slot] = fc;
                }
                __syncthreads();
            }
        }
    }
    for (int e = (int)threadIdx.x; e < n; e += (int)blockDim.x) a.gbest[e] = G[e];
    if (threadIdx.x == 0) publish(st, it, gfit, status, a.maxiter);
}

typedef void (*de_async_t)(const sx_de_args, const PlanArg);
typedef void (*pso_async_t)(const sx_pso_args, const PlanArg);

template <int RNG, int LPR, int NFIX = 0>
de_async_t pick_de(int fun_id) {
    switch (fun_id) {
        case SX_FUN_ACKLEY: return de_async_kernel<SX_FUN_ACKLEY, RNG, LPR, NFIX>;
        case SX_FUN_RASTRIGIN: return de_async_kernel<SX_FUN_RASTRIGIN, RNG, LPR, NFIX>;
        case SX_FUN_ROSENBROCK: return de_async_kernel<SX_FUN_ROSENBROCK, RNG, LPR, NFIX>;
        case SX_FUN_SPHERE: return de_async_kernel<SX_FUN_SPHERE, RNG, LPR, NFIX>;
        // (the others: the run-time row length only -- hot_objective, sx_device.hpp)
        case SX_FUN_GRIEWANK: return de_async_kernel<SX_FUN_GRIEWANK, RNG, LPR, 0>;
        case SX_FUN_QUARTIC: return de_async_kernel<SX_FUN_QUARTIC, RNG, LPR, 0>;
        case SX_FUN_STYBLINSKI_TANG: return de_async_kernel<SX_FUN_STYBLINSKI_TANG, RNG, LPR, 0>;
    }
    return nullptr;
}
template <int RNG, int LPR, int NFIX = 0>
pso_async_t pick_pso(int fun_id) {
    switch (fun_id) {
        case SX_FUN_ACKLEY: return pso_async_kernel<SX_FUN_ACKLEY, RNG, LPR, NFIX>;
        case SX_FUN_RASTRIGIN: return pso_async_kernel<SX_FUN_RASTRIGIN, RNG, LPR, NFIX>;
        case SX_FUN_ROSENBROCK: return pso_async_kernel<SX_FUN_ROSENBROCK, RNG, LPR, NFIX>;
        case SX_FUN_SPHERE: return pso_async_kernel<SX_FUN_SPHERE, RNG, LPR, NFIX>;
        // (the others: the run-time row length only -- hot_objective, sx_device.hpp)
        case SX_FUN_GRIEWANK: return pso_async_kernel<SX_FUN_GRIEWANK, RNG, LPR, 0>;
        case SX_FUN_QUARTIC: return pso_async_kernel<SX_FUN_QUARTIC, RNG, LPR, 0>;
        case SX_FUN_STYBLINSKI_TANG: return pso_async_kernel<SX_FUN_STYBLINSKI_TANG, RNG, LPR, 0>;
    }
    return nullptr;
}

template <typename K>
int launch_sweep(K kern, const SweepGeometry &g, hipStream_t s, const void *args_struct) {
    (void)args_struct;
    // more than the default 64 KiB of dynamic LDS: opt in per kernel (160 KB per CU on gfx950)
    SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)g.lds));
    return 0;
}

}  // namespace

#if SX_ASYNC_PART == 0
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
        SX_DISPATCH_LPR(a->n, kern = a->n == 4 * LPR ? (pick_de<SX_RNG_PHILOX, LPR, 4 * LPR>(a->fun_id))
                                                      : (pick_de<SX_RNG_PHILOX, LPR>(a->fun_id)))
    } else {
        SX_DISPATCH_LPR(a->n, kern = (pick_de<SX_RNG_HOST, LPR>(a->fun_id)))
    }
    const SweepGeometry g = sweep_geometry(a->n, 0, a->fun_id);
    if (int rc = launch_sweep(kern, g, (hipStream_t)stream, a)) return rc;
    hipLaunchKernelGGL(kern, dim3(1), dim3(g.waves * kWave), g.lds, (hipStream_t)stream, *a, plan);
    SX_LAUNCH_CHECK();
    return 0;
}

#else
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
        SX_DISPATCH_LPR(a->n, kern = a->n == 4 * LPR ? (pick_pso<SX_RNG_PHILOX, LPR, 4 * LPR>(a->fun_id))
                                                      : (pick_pso<SX_RNG_PHILOX, LPR>(a->fun_id)))
    } else {
        SX_DISPATCH_LPR(a->n, kern = (pick_pso<SX_RNG_HOST, LPR>(a->fun_id)))
    }
    const SweepGeometry g = sweep_geometry(a->n, 1, a->fun_id);
    if (int rc = launch_sweep(kern, g, (hipStream_t)stream, a)) return rc;
    hipLaunchKernelGGL(kern, dim3(1), dim3(g.waves * kWave), g.lds, (hipStream_t)stream, *a, plan);
    SX_LAUNCH_CHECK();
    return 0;
}
#endif
