// PSO / CPSO: one fused kernel per generation (velocity + position update, Shrink
// clamp, objective, personal-best selection, per-workgroup best) and the
// competitive-restart kernels (swarm radius, worst-nw selection, re-seeding).
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/cpso/_cpso.py:324-329   mutation (left-to-right association)
//   stochopy/optimize/cpso/_cpso.py:332-361   pso_sync
//   stochopy/optimize/cpso/_constraints.py:4-10, 44-53  NoConstraint / Shrink (sync form)
//   stochopy/optimize/cpso/_cpso.py:405-426   restart (radius, nw, worst-nw reset)
//   stochopy/optimize/_common.py:123-130      selection (strict <, in place)
//   stochopy/factory/benchmark.py             objective, fused
//
// One wavefront per particle; X, V, pbest, pbestfit are row-local and updated in place.
#include <cstdlib>

#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"
#include "sx_wide.hpp"

namespace sx {
int make_plan_arg(int fun_id, int n, PlanArg *out);
}
using namespace sx;

namespace {

// order-preserving map double -> uint64 (larger double <=> larger key)
__device__ __forceinline__ unsigned long long sort_key(double f) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(f);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// FULL: n == 4 * LPR (64, 128 or 256 -- BASELINE config 3): the row length is a compile-time constant, a row is exactly
// one batch, and every bound check and loop over the row folds away.
// PLAIN (round 5: with or without FULL): constraints=None and no restart pending (plain PSO, or the first generation of a CPSO graph): neither
// the Shrink pass nor the re-seeding test is compiled in.
// CHAIN (with PLAIN; round 3): ONE kernel per generation, the way DE runs (sx_de_kernel.hpp).  Launch L (parity
// chain_p = L & 1) first finalises the generation its predecessor produced -- every workgroup reduces the per-workgroup
// records of parity chain_p (two levels: a slice per thread, DPP over the wave, the waves through LDS), derives the same
// best / status, workgroup 0 publishes it in state[1 - chain_p] -- and then produces the next generation, writing its
// record of parity 1 - chain_p.  The swarm is updated in place, so the best row cannot be read from pbest while other
// workgroups overwrite theirs: every workgroup keeps a copy of ITS best row in best_rows[q][block] and its record says
// which q (rec = 2 * row + q); the copy is rewritten -- into the buffer the current record does NOT point to -- only
// when the workgroup's best row changed.  mode 1: one workgroup, finalise only, into state[2] (the host's view).
constexpr int kChainRecPerThread = 8;

// ONEB (round 5): a whole-wave kernel instantiated for rows of 129 ... 256 elements of run-time length: one batch, the objective's
// run-time register chain only -- the general whole-wave kernel carries the long rows' summation plans in its register budget
// (111 VGPRs with Ackley: two workgroups per CU, where the n = 256 kernel runs four)
#ifndef SX_PSO_RAD_WAVES
#define SX_PSO_RAD_WAVES 6
#endif
template <int FUN, int RNG, int LPR, bool FULL, bool PLAIN = false, bool CHAIN = false, bool ONEB = false>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave, (FULL && !PLAIN && !CHAIN && LPR == kWave) ? SX_PSO_RAD_WAVES : 1) void pso_generation_kernel(const sx_pso_args a,
                                                                                const PlanArg plan,
                                                                                double *__restrict__ best_rows,
                                                                                const int chain_p, const int mode,
                                                                                const int64_t npart) {
    static_assert(!CHAIN || (PLAIN && FULL), "the chained form exists for the whole-batch constraints=None kernel");
    // RAD (CPSO inside a graph; best_rows then points at npart doubles, see sx_pso_graph_create): the workgroup also leaves
    // max_i ||X_i(new) - gbest(OLD)||_2^2 over its rows -- the (squared) swarm radius against the best the kernel was started with, in
    // pso_radius_kernel's own order of operations.  cpso_post_kernel turns that into the restart decision (radius_decision)
    // and the radius pass over X is skipped.
    constexpr bool RAD = FULL && !PLAIN && !CHAIN;
    double racc = 0.0;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sf[kMaxRowsPerBlock];
    __shared__ int64_t si[kMaxRowsPerBlock];
    __shared__ int sb[kMaxRowsPerBlock];          // CHAIN: the row's pbest changed in this generation
    __shared__ double s_wf[kMaxWavesPerBlock];    // CHAIN: the waves' partial results of the record reduction
    __shared__ int64_t s_wi[kMaxWavesPerBlock];
    // the state word (a miss after every kernel boundary) is needed by the Philox counters and the stop test only: the
    // PLAIN kernel issues the row loads of its batch before anything waits for it.  (Not the general one: with the
    // re-seeding code behind it the late test costs 130 VGPRs instead of 80 -- 3 waves per SIMD instead of 6 -- and made
    // that kernel 31.5 -> 53 us, profiles/r2_pso_c3_variants.txt.)
    const sx_state *st = CHAIN ? a.state + chain_p : a.state;
    const int done = st->done;
    if (!PLAIN && done) return;
    // CHAIN: st->it is the last FINALISED generation; the swarm holds st->it + 1, this launch produces st->it + 2
    const int64_t it_held = CHAIN ? st->it + 1 : st->it;
    const uint32_t gen = (uint32_t)(it_held + 1);
    const int n = FULL ? 4 * LPR : a.n;
    const int64_t P = a.P, ld = a.ld;
    const RowIds<LPR> id(P);
    const int l = id.l;  // lane within the row
    const int64_t rowc = id.rowc;
    double *U = lds + id.slot * gen_row_stride(n);
    double *Vn = U;  // Shrink: the raw velocity waits in U[e] until the owning lane replaces it by the position

    double fold = a.pbestfit[rowc];
    // CPSO inside a graph: the restart decided at the end of the previous generation is carried out here -- a selected
    // row is not loaded but re-seeded (what pso_restart_apply_kernel would have written: same Philox positions, V = 0,
    // pbest = X, pbestfit = 1e30), and then moved like any other
    bool reseed = false;
    if (!PLAIN && RNG == SX_RNG_PHILOX && a.pending_restart != nullptr)
        reseed = a.pending_restart[0] != 0ull && sort_key(fold) >= a.pending_restart[1];
    if (reseed) fold = 1.0e30;
    double *__restrict__ xr = a.X + rowc * ld;
    double *__restrict__ vr = a.V + rowc * ld;
    double *__restrict__ pb = a.pbest + rowc * ld;
    const double *__restrict__ gb = a.gbest;
    // CHAIN: the predecessor's records, a contiguous slice per thread (first-minimum rule), and this workgroup's own
    double pfv[kChainRecPerThread];
    int64_t piv[kChainRecPerThread];
    int64_t myprev = 0;
    double *part_f_out = a.part_f;
    int64_t *part_i_out = a.part_i;
    if (CHAIN) {
        const double *pf = a.part_f + (int64_t)chain_p * npart;
        const int64_t *pi = a.part_i + (int64_t)chain_p * npart;
        const int per = (int)((npart + blockDim.x - 1) / blockDim.x);
        const int64_t k0 = (int64_t)threadIdx.x * per;
#pragma unroll
        for (int u = 0; u < kChainRecPerThread; ++u) {
            const bool in = u < per && k0 + u < npart;
            pfv[u] = in ? pf[k0 + u] : __builtin_huge_val();
            piv[u] = in ? pi[k0 + u] : INT64_MAX;
        }
        myprev = pi[id.block];
        part_f_out = a.part_f + (int64_t)(1 - chain_p) * npart;
        part_i_out = a.part_i + (int64_t)(1 - chain_p) * npart;
    }
    const uint32_t grow = (uint32_t)(a.row0 + rowc);
    const double w = a.w, c1 = a.c1, c2 = a.c2;
    const bool shrink = PLAIN ? false : a.constraints != 0;
    const double *r1row = RNG == SX_RNG_HOST ? a.r1 + rowc * (int64_t)n : nullptr;
    const double *r2row = RNG == SX_RNG_HOST ? a.r2 + rowc * (int64_t)n : nullptr;

    // V = w*V + c1*r1*(pbest - X) + c2*r2*(gbest - X)   (cpso/_cpso.py:326), four row steps per batch so
    // the 16 loads of a batch are in flight together.  Without Shrink the new position follows at once;
    // with Shrink the raw velocity waits in LDS for the row-wide beta.
    constexpr int kStep = 4;
    double beta = __builtin_huge_val();
    const int nq = (ONEB || LPR < kWave) ? kStep : (n + LPR - 1) / LPR;  // (short rows and ONEB rows: one batch, said at compile time)
    for (int q0 = 0; q0 < nq; q0 += kStep) {
        double x[kStep], v[kStep], p[kStep], g[kStep], r1[kStep], r2[kStep];
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            const bool in = FULL || e < n, ld_row = in && !reseed;
            x[t] = ld_row ? xr[e] : 0.0;
            v[t] = ld_row ? vr[e] : 0.0;
            p[t] = ld_row ? pb[e] : 0.0;
            if (!CHAIN) g[t] = in ? gb[e] : 0.0;
            r1[t] = (RNG == SX_RNG_HOST && in) ? r1row[e] : 0.0;
            r2[t] = (RNG == SX_RNG_HOST && in) ? r2row[e] : 0.0;
        }
        if (PLAIN && done) {  // (uniform; nothing has been written yet)
            if (CHAIN && mode == 1 && blockIdx.x == 0 && threadIdx.x == 0) a.state[2] = *st;
            return;
        }
        if constexpr (CHAIN) {
            // ---- finalise the generation the swarm holds: (min f, first row) over the records, two levels ----
            double bf = pfv[0];
            int64_t brec = piv[0];
#pragma unroll
            for (int u = 1; u < kChainRecPerThread; ++u)
                if (pfv[u] < bf) bf = pfv[u], brec = piv[u];  // (slices are in row order: strict < keeps the first)
            wave_argmin_ordered(bf, brec);
            if (id.lane == 0) s_wf[id.wave] = bf, s_wi[id.wave] = brec;
            __syncthreads();
            bf = s_wf[0], brec = s_wi[0];
            const int nw = (int)(blockDim.x >> 6);
            for (int wv = 1; wv < nw; ++wv)
                if (s_wf[wv] < bf) bf = s_wf[wv], brec = s_wi[wv];
            int status = SX_STATUS_NONE;
            if (it_held >= 2) {  // the reference does not test the initial swarm (cpso/_cpso.py:219-240)
                if (bf <= a.ftol)
                    status = 1;  // (0 if the best moved by <= xtol: settled by the host from the two resident rows)
                else if (it_held >= a.maxiter)
                    status = -1;
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                sx_state *so = a.state + (mode == 1 ? 2 : 1 - chain_p);
                so->it = it_held;
                so->gbidx = brec >> 1;
                so->gfit = bf;
                so->dx = 0.0;
                so->status = status;
                so->done = status != SX_STATUS_NONE;
                so->reserved[0] = st->reserved[1];  // the record of the best of the generation before
                so->reserved[1] = brec;
            }
            if (status != SX_STATUS_NONE || mode == 1) return;
            const int rows_in_block = (int)(blockDim.x >> 6) * RowIds<LPR>::RPW;
            const double *__restrict__ gbc = best_rows + ((brec & 1) * npart + (brec >> 1) / rows_in_block) * (int64_t)n;
#pragma unroll
            for (int t = 0; t < kStep; ++t) g[t] = gbc[(q0 + t) * LPR + l];
        }
        if (RNG == SX_RNG_PHILOX) {
            // 53-bit r1 and r2, as the reference's two rand(P, n) blocks (cpso/_cpso.py:262-263): a call per purpose and pair
            // of steps (slot = (q>>1)*LPR + l, half = q&1).  (Rounds 1-5: 32-bit ones, one call for both -- +2.8 % at C3a / C3b
            // for the reference's own precision, profiles/r6_philox53.txt.)
#pragma unroll
            for (int t = 0; t < kStep; t += 2) {
                const uint32_t slot = (uint32_t)((q0 + t) >> 1) * (uint32_t)LPR + (uint32_t)l;
                const U4 wa = philox4x32_10(slot, grow, gen, kPurposePsoR1, a.key0, a.key1);
                const U4 wb = philox4x32_10(slot, grow, gen, kPurposePsoR2, a.key0, a.key1);
                r1[t] = u53(wa.x, wa.y);
                r1[t + 1] = u53(wa.z, wa.w);
                r2[t] = u53(wb.x, wb.y);
                r2[t + 1] = u53(wb.z, wb.w);
            }
        }
        if (RNG == SX_RNG_PHILOX && reseed) {  // X = uniform(lower, upper) keyed like pso_restart_apply_kernel's draws
#pragma unroll
            for (int t = 0; t < kStep; t += 2) {
                const U4 wd = philox4x32_10((uint32_t)((q0 + t) >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen - 1u,
                                            kPurposePsoRestart, a.key0, a.key1);
                const double u[2] = {u53(wd.x, wd.y), u53(wd.z, wd.w)};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = (q0 + t + h) * LPR + l;
                    if (e < n) {
                        const double lo = a.lower[e];
                        x[t + h] = lo + (a.upper[e] - lo) * u[h];
                        p[t + h] = x[t + h];
                        // pbest = X now (a row whose new fitness is not below 1e30 keeps it); Shrink's second pass
                        // reads X back
                        if (id.active) {
                            pb[e] = x[t + h];
                            if (shrink) xr[e] = x[t + h];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (e < n) {
                const double vn = pso_velocity(w, v[t], c1, r1[t], p[t], x[t], c2, r2[t], g[t]);
                if (shrink) {  // cpso/_constraints.py:22-50: beta = min over violated dims of (bound - x)/v
                    Vn[e] = vn;
                    const double xc = x[t] + vn;
                    const double lo = a.lower[e], hi = a.upper[e];
                    if (xc < lo) beta = fmin(beta, (lo - x[t]) / vn);
                    if (xc > hi) beta = fmin(beta, (hi - x[t]) / vn);
                } else {  // cpso/_constraints.py:4-10: X + V
                    const double xn = x[t] + vn;
                    if (RAD) {
                        const double d = xn - g[t];
                        racc += d * d;
                    }
                    U[e] = xn;
                    if (id.active) {
                        vr[e] = vn;
                        xr[e] = xn;
                    }
                }
            }
        }
    }
    if (shrink) {
        beta = row_min<LPR>(beta);
        if (beta == __builtin_huge_val()) beta = 1.0;
        lds_wave_fence();
        for (int e = l; e < n; e += LPR) {
            const double vn = Vn[e] * beta;  // V *= beta[:, None]
            const double xn = xr[e] + vn;
            if (RAD) {
                const double d = xn - gb[e];
                racc += d * d;
            }
            U[e] = xn;
            if (id.active) {
                vr[e] = vn;
                xr[e] = xn;
            }
        }
    }
    const double fc = row_objective<FUN, LPR, FULL, FULL ? 4 * LPR : 0, SX_LONG_STATIC, ONEB>(U, n, plan, l);
    const bool better = fc < fold;  // _common.py:127 strict <
    if (id.active) {
        if (better)
            for (int e = l; e < n; e += LPR) pb[e] = U[e];
        if (l == 0) {
            if (better)
                a.pbestfit[id.row] = fc;
            else if (reseed)
                a.pbestfit[id.row] = 1.0e30;
            if (a.candfit != nullptr) a.candfit[id.row] = fc;
        }
    }
    if constexpr (!CHAIN) {
        __shared__ double sr[kMaxRowsPerBlock];
        const bool rad = RAD && best_rows != nullptr;  // (uniform) exactly pso_radius_kernel's reduction, behind the records' barrier
        if (rad) {
            // the SQUARED radius (the square root is monotone and correctly rounded: max sqrt = sqrt max, taken once by
            // cpso_post_kernel); whole-wave rows add up without LDS traffic (same additions as row_sum)
            if constexpr (LPR == kWave)
                racc = wave_sum_butterfly(racc, id.lane);
            else
                racc = row_sum<LPR>(racc);
            if (id.l == 0) {
                sr[id.slot] = id.active ? racc : 0.0;
                if (id.active) best_rows[npart + id.row] = racc;  // (per row, behind the npart per-workgroup maxima)
            }
        }
        block_partial<LPR>(better ? fc : fold, id, sf, si, a.part_f, a.part_i);
        if (rad && threadIdx.x < kWave) {
            const int rows_in_block = (int)(blockDim.x >> 6) * RowIds<LPR>::RPW;
            const double m = wave_max_f64((int)threadIdx.x < rows_in_block ? sr[threadIdx.x] : 0.0);
            if (threadIdx.x == 0) best_rows[blockIdx.x] = m;
        }
    } else {
        // the workgroup's record AND, when its best row changed, the row itself (see the head of the kernel)
        if (id.l == 0) {
            sf[id.slot] = id.active ? (better ? fc : fold) : __builtin_huge_val();
            si[id.slot] = id.active ? id.row : INT64_MAX;
            sb[id.slot] = better ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x < kWave) {
            const int rows_in_block = (int)(blockDim.x >> 6) * RowIds<LPR>::RPW;
            const int k = (int)threadIdx.x;
            double bf = k < rows_in_block ? sf[k] : __builtin_huge_val();
            int64_t bi = k < rows_in_block ? si[k] : INT64_MAX;
            wave_argmin_ordered(bf, bi);
            const int ks = (int)(bi - (int64_t)id.block * rows_in_block);  // the winning slot
            const bool changed = sb[ks] != 0 || bi != (myprev >> 1);
            const int64_t q = changed ? 1 - (myprev & 1) : (myprev & 1);
            if (changed) {  // the new pbest of a row that improved is its position, still in LDS
                const double *src = sb[ks] ? lds + ks * gen_row_stride(n) : a.pbest + bi * ld;
                double *dst = best_rows + (q * npart + id.block) * (int64_t)n;
                for (int e = k; e < n; e += kWave) dst[e] = src[e];
            }
            if (k == 0) {
                part_f_out[id.block] = bf;
                part_i_out[id.block] = 2 * bi + q;
            }
        }
    }
}

typedef void (*pso_kernel_t)(const sx_pso_args, const PlanArg, double *, int, int, int64_t);

template <int RNG, int LPR, bool FULL, bool PLAIN = false, bool CHAIN = false, bool ONEB = false>
pso_kernel_t pick_kernel_lpr(int fun_id) {
    constexpr bool SPECIAL = PLAIN || ONEB || CHAIN;  // (pick_kernel / sx_pso_chain_supported route the other objectives to the general form)
    switch (fun_id) {
        case SX_FUN_ACKLEY: return pso_generation_kernel<SX_FUN_ACKLEY, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
        case SX_FUN_RASTRIGIN: return pso_generation_kernel<SX_FUN_RASTRIGIN, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
        case SX_FUN_ROSENBROCK: return pso_generation_kernel<SX_FUN_ROSENBROCK, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
        case SX_FUN_SPHERE: return pso_generation_kernel<SX_FUN_SPHERE, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
    }
    if constexpr (!SPECIAL) {
        switch (fun_id) {
            case SX_FUN_GRIEWANK: return pso_generation_kernel<SX_FUN_GRIEWANK, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
            case SX_FUN_QUARTIC: return pso_generation_kernel<SX_FUN_QUARTIC, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
            case SX_FUN_STYBLINSKI_TANG: return pso_generation_kernel<SX_FUN_STYBLINSKI_TANG, RNG, LPR, FULL, PLAIN, CHAIN, ONEB>;
        }
    }
    return nullptr;
}

// the chained form: whole-batch rows (n = 64, 128, 256), in-kernel draws, constraints=None, no restart
pso_kernel_t pick_chain_kernel(int fun_id, int n) {
    switch (lanes_per_row(n)) {
        case 16: return pick_kernel_lpr<SX_RNG_PHILOX, 16, true, true, true>(fun_id);
        case 32: return pick_kernel_lpr<SX_RNG_PHILOX, 32, true, true, true>(fun_id);
    }
    return pick_kernel_lpr<SX_RNG_PHILOX, 64, true, true, true>(fun_id);
}

template <int RNG>
pso_kernel_t pick_kernel(int fun_id, int n, bool plain) {
    constexpr bool PH = RNG == SX_RNG_PHILOX;
    const bool hot = hot_objective(fun_id);
    plain = plain && hot;  // (PLAIN and the one-batch form: the four hot objectives; the others take the general kernel)
    const int lpr = lanes_per_row(n);
    // whole-batch rows with in-kernel draws get the constant-length form (host draws: the run is bound by the host)
    const bool full = PH && n == 4 * lpr;
    // round 5: rows of any other length get the PLAIN form too when nothing needs the general one (constraints=None, no
    // restart pending): without the Shrink pass and the re-seeding code the kernel runs at 8 waves per SIMD instead of 6 --
    // plain PSO at n = 250 / 300 was 0.75 / 0.67 of its n = 256 neighbour's HBM fraction (profiles/r5_shapes_before.txt)
    const bool pl = PH && plain;
    switch (lpr) {
        case 16:
            if (full) return plain ? pick_kernel_lpr<RNG, 16, PH, PH>(fun_id) : pick_kernel_lpr<RNG, 16, PH>(fun_id);
            return pl ? pick_kernel_lpr<RNG, 16, false, PH>(fun_id) : pick_kernel_lpr<RNG, 16, false>(fun_id);
        case 32:
            if (full) return plain ? pick_kernel_lpr<RNG, 32, PH, PH>(fun_id) : pick_kernel_lpr<RNG, 32, PH>(fun_id);
            return pl ? pick_kernel_lpr<RNG, 32, false, PH>(fun_id) : pick_kernel_lpr<RNG, 32, false>(fun_id);
    }
    if (full) return plain ? pick_kernel_lpr<RNG, 64, PH, PH>(fun_id) : pick_kernel_lpr<RNG, 64, PH>(fun_id);
    if (PH && n <= 4 * kWave && hot)  // rows of 129 ... 256 elements off the grid: the one-batch form of the whole-wave kernel
        return pl ? pick_kernel_lpr<RNG, 64, false, PH, false, PH>(fun_id) : pick_kernel_lpr<RNG, 64, false, false, false, PH>(fun_id);
    return pl ? pick_kernel_lpr<RNG, 64, false, PH>(fun_id) : pick_kernel_lpr<RNG, 64, false>(fun_id);
}

int check_args(const sx_pso_args *a) {
    SX_REQUIRE(a != nullptr, "sx_pso: null args");
    SX_REQUIRE(a->X && a->V && a->pbest && a->pbestfit && a->gbest && a->state && a->part_f && a->part_i,
               "sx_pso: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->n >= 1 && a->ld >= a->n, "sx_pso: bad shape");
    SX_REQUIRE(a->fun_id >= 0 && a->fun_id < SX_FUN_COUNT, "sx_pso: unknown objective");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_pso: unknown rng mode");
    SX_REQUIRE(a->rng != SX_RNG_HOST || (a->r1 && a->r2), "sx_pso: host draws missing");
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper), "sx_pso: bounds missing");
    return 0;
}

Geometry geometry(int64_t P, int n) {
    Geometry g = row_geometry(P, n);
    // rows of up to 256 elements stage nothing but the new position (sx_device.hpp gen_row_stride)
    if (!is_wide(n)) g.lds = (size_t)rows_per_block(n) * gen_row_stride(n) * sizeof(double);
    return g;
}

// ---------------------------------------------------------------------------
// Competitive restart, cpso/_cpso.py:405-426
// ---------------------------------------------------------------------------
// per-workgroup max_i ||X_i - gbest||_2  (:410)
template <int LPR>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave) void pso_radius_kernel(const sx_pso_args a,
                                                                              double *__restrict__ part_r) {
    __shared__ double sr[kMaxRowsPerBlock];
    const int done = a.state->done;  // looked at behind the row loads (a miss after the kernel boundary)
    const RowIds<LPR> id(a.P);
    const double *__restrict__ xr = a.X + id.rowc * a.ld;
    double acc = 0.0;
    for (int e0 = id.l; e0 < a.n; e0 += 4 * LPR) {  // four row loads in flight per lane
        double xv[4], gv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = e0 + t * LPR;
            xv[t] = e < a.n ? xr[e] : 0.0;
            gv[t] = e < a.n ? a.gbest[e] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double d = xv[t] - gv[t];
            acc += d * d;
        }
    }
    if (done) return;
    acc = sqrt(row_sum<LPR>(acc));
    if (id.l == 0) sr[id.slot] = id.active ? acc : 0.0;
    __syncthreads();
    if (threadIdx.x < kWave) {  // wavefront 0: slot k in lane k
        const int rows_in_block = (int)(blockDim.x >> 6) * RowIds<LPR>::RPW;
        const double m = wave_max_f64((int)threadIdx.x < rows_in_block ? sr[threadIdx.x] : 0.0);
        if (threadIdx.x == 0) part_r[blockIdx.x] = m;
    }
}

// histogram increment for one key per lane (dig < 0: this lane has none).  Fitness values of a converged swarm
// pile up in very few bins, where plain LDS atomics would serialise: the wave first looks for large groups of
// equal digits (up to four) and adds one count per group; whatever is left is spread out and goes as plain atomics.
__device__ __forceinline__ void hist_add(unsigned *bins, int dig, int lane) {
    unsigned long long todo = __ballot(dig >= 0);
    for (int r = 0; r < 4 && todo; ++r) {
        const int leader = (int)__ffsll((long long)todo) - 1;
        const int d = __shfl(dig, leader, kWave);
        const unsigned long long same = __ballot(dig == d);
        if (__popcll(same) < 4) break;
        if (lane == leader) atomicAdd(&bins[d], (unsigned)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&bins[dig], 1u);
}

constexpr int kSelThreads = 1024;
constexpr int kSelPerThread = 16;  // keys held in registers up to 16384 particles (larger swarms re-read them each step)

#ifdef SX_SELTRACE
#define SEL_TP(k) do { if (threadIdx.x == 0) ((unsigned long long *)a.candfit)[k] = wall_clock64(); } while (0)
#else
#define SEL_TP(k) do {} while (0)
#endif

// The nw-th largest pbestfit key of the (segmented) swarm -> out[1]; every thread of the 1024-thread workgroup takes part.
// (The second half of pso_restart_select_kernel; cpso_post_kernel runs it behind its own restart decision.)
__device__ __forceinline__ void restart_threshold(const sx_pso_args &a, const double *__restrict__ fit, const int nseg,
                                                  const int64_t seg_len, const int64_t seg_stride, const int64_t Ptot,
                                                  const int64_t nw, unsigned long long *__restrict__ out) {
    const unsigned useg = (unsigned)seg_len;
    auto fit_at = [&](int64_t i) -> double {
        if (nseg == 1) return fit[i];
        const unsigned sg = (unsigned)i / useg;
        return fit[(int64_t)sg * seg_stride + ((unsigned)i - sg * useg)];
    };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // up to 32768 particles the keys stay in registers for the 8 passes; larger swarms re-read them (L2)
    const bool in_regs = Ptot <= (int64_t)kSelThreads * kSelPerThread;
    unsigned long long key[kSelPerThread];
    // (the loops over the register array are fully unrolled; `k * kSelThreads < Ptot` is uniform, so the
    //  iterations past the swarm cost one scalar branch each)
    // loads in batches of 8, unconditionally (indices past the swarm are clamped): one latency per batch, not per load
#pragma unroll
    for (int k0 = 0; k0 < kSelPerThread; k0 += 8) {
        double fv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = (int64_t)(k0 + u) * kSelThreads + tid;
            fv[u] = (in_regs && (int64_t)k0 * kSelThreads < Ptot) ? fit_at(i < Ptot ? i : Ptot - 1) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = (int64_t)(k0 + u) * kSelThreads + tid;
            key[k0 + u] = (in_regs && i < Ptot) ? sort_key(fv[u]) : 0ull;  // 0 < real keys
        }
    }
    SEL_TP(2);
    // 0. (round 3) A histogram over ALL keys is bound by the LDS atomic rate -- 16 384 atomics at about one per cycle are
    //    the 6.5-8.8 us the step took at BASELINE config 3b, whatever the digits look like (tools/seltrace.py).  So the
    //    keys are first cut down to a WINDOW around the target: 128 of them (the first key of the first 128 threads) are
    //    ranked among themselves (eight threads per sample, 16 comparisons each); the target's rank nw, scaled to the
    //    sample, +- kWinHalf sample ranks (+- 2.5 sigma of the sampling error at the median) gives two keys lo <= hi; one
    //    pass over the register-held keys counts those above hi and packs those inside [lo, hi] into LDS (one returning
    //    atomic per WAVE).  If the target lies inside -- nw - above in [1, window size], the window fits -- the selection
    //    goes on over the window (a few thousand keys) exactly as it would over the whole swarm; otherwise over the whole
    //    swarm, as before.  Exact either way: the k-th largest of all keys is the (k - above)-th largest of the window.
    constexpr int kSample = 128, kWinHalf = 14, kWinCap = 4096;
    __shared__ unsigned long long s_sample[kSample], s_win[kWinCap], s_lohi[2];
    __shared__ unsigned s_nwin, s_abv[kSelThreads / kWave];
    bool use_regs = in_regs;      // where the keys of the steps below come from: registers, the window, or memory
    int64_t Peff = Ptot;          // how many there are
    unsigned remaining0 = (unsigned)nw;
    const bool try_window = in_regs && Ptot >= 2 * kWinCap;
    if (try_window) {
        if (tid < kSample) s_sample[tid] = key[0];  // (Ptot >= 16384: the first 256 keys exist)
        if (tid == 0) s_nwin = 0u, s_lohi[0] = 0ull, s_lohi[1] = ~0ull;
        __syncthreads();
        {   // descending rank of sample (tid >> 3): lanes 8s .. 8s+7 take an eighth of the comparisons each
            const unsigned long long mine = s_sample[tid >> 3];
            unsigned g = 0u, e = 0u;
            const int q0 = (tid & 7) * (kSample / 8);
#pragma unroll
            for (int c = q0; c < q0 + kSample / 8; ++c) {
                const unsigned long long o = s_sample[c];
                g += o > mine;
                e += o == mine;
            }
            g += __shfl_xor(g, 1, kWave), e += __shfl_xor(e, 1, kWave);
            g += __shfl_xor(g, 2, kWave), e += __shfl_xor(e, 2, kWave);
            g += __shfl_xor(g, 4, kWave), e += __shfl_xor(e, 4, kWave);
            // target rank nw of Ptot  ->  sample rank r (1-based, descending); the window's ends are the samples at ranks
            // r - kWinHalf (hi) and r + kWinHalf (lo), or the extremes of the key range beyond the sample's ends
            const int r = (int)(((int64_t)nw * kSample + Ptot - 1) / Ptot);
            const int rhi = r - kWinHalf, rlo = r + kWinHalf;
            if ((tid & 7) == 0) {
                if (rhi >= 1 && (int)g < rhi && rhi <= (int)(g + e)) s_lohi[1] = mine;
                if (rlo <= kSample && (int)g < rlo && rlo <= (int)(g + e)) s_lohi[0] = mine;
            }
        }
        __syncthreads();
        const unsigned long long lo = s_lohi[0], hi = s_lohi[1];
        unsigned above = 0u, inmask = 0u;  // (bit k: key k lies inside the window -- one comparison pass, the packing walks the bits)
#pragma unroll
        for (int k = 0; k < kSelPerThread; ++k) {
            const int64_t i = (int64_t)k * kSelThreads + tid;
            const bool valid = (int64_t)k * kSelThreads < Ptot && i < Ptot;
            above += valid && key[k] > hi;
            inmask |= (valid && key[k] >= lo && key[k] <= hi) ? (1u << k) : 0u;
        }
        const unsigned mine_in = (unsigned)__popc(inmask);
        // exclusive prefix of mine_in over the wave + the wave's base in the window (one returning LDS atomic per wave)
        unsigned incl = mine_in;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        unsigned wbase = 0u;
        const unsigned wtot = __shfl(incl, kWave - 1, kWave);
        if (lane == kWave - 1) wbase = atomicAdd(&s_nwin, wtot);
        wbase = __shfl(wbase, kWave - 1, kWave);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) above += __shfl_xor(above, off, kWave);
        if (lane == 0) s_abv[wv] = above;
        unsigned pos = wbase + incl - mine_in;
#pragma unroll
        for (int k = 0; k < kSelPerThread; ++k) {
            if ((inmask >> k) & 1u) {
                if (pos < (unsigned)kWinCap) s_win[pos] = key[k];
                ++pos;
            }
        }
        __syncthreads();
        unsigned abv = 0u;
#pragma unroll
        for (int k = 0; k < kSelThreads / kWave; ++k) abv += s_abv[k];
        const unsigned nwin = s_nwin;
        if (nwin <= (unsigned)kWinCap && (unsigned)nw > abv && (unsigned)nw - abv <= nwin) {
            use_regs = false;
            Peff = (int64_t)nwin;
            remaining0 = (unsigned)nw - abv;
        }
    }
    const bool from_window = try_window && !use_regs;
    // 1. the keys' common leading bits carry no information (sign, exponent and the first mantissa bits are the
    //    same all over a converged swarm): find the highest bit in which any two keys differ
    __shared__ unsigned long long s_min[kSelThreads / kWave], s_max[kSelThreads / kWave];
    unsigned long long kmin = ~0ull, kmax = 0ull;
    if (use_regs) {
#pragma unroll
        for (int k = 0; k < kSelPerThread; ++k) {
            const int64_t i = (int64_t)k * kSelThreads + tid;
            if ((int64_t)k * kSelThreads < Ptot && i < Ptot) {
                kmin = key[k] < kmin ? key[k] : kmin;
                kmax = key[k] > kmax ? key[k] : kmax;
            }
        }
    } else {
        if (from_window) {
            for (int i = tid; i < (int)Peff; i += kSelThreads) {
                const unsigned long long kk = s_win[i];
                kmin = kk < kmin ? kk : kmin;
                kmax = kk > kmax ? kk : kmax;
            }
        } else {
            for (int64_t i = tid; i < Peff; i += kSelThreads) {
                const unsigned long long kk = sort_key(fit_at(i));
                kmin = kk < kmin ? kk : kmin;
                kmax = kk > kmax ? kk : kmax;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long omin = __shfl_xor(kmin, off, kWave), omax = __shfl_xor(kmax, off, kWave);
        kmin = omin < kmin ? omin : kmin;
        kmax = omax > kmax ? omax : kmax;
    }
    if (lane == 0) {
        s_min[wv] = kmin;
        s_max[wv] = kmax;
    }
    __syncthreads();
    for (int k = 0; k < kSelThreads / kWave; ++k) {
        kmin = s_min[k] < kmin ? s_min[k] : kmin;
        kmax = s_max[k] > kmax ? s_max[k] : kmax;
    }
    SEL_TP(3);
    if (kmin == kmax) {  // one value all over the swarm: it is the threshold
        if (tid == 0) out[1] = kmin;
        return;
    }
    // 2. radix descent from that bit, up to 10 bits per step (one histogram bin per thread): histogram (LDS
    //    atomics) of the digit over the keys that match the prefix found so far; the digit of the `remaining`-th
    //    largest of them is where the suffix count crosses it.  As soon as at most 512 keys share the prefix --
    //    after the first step, as a rule -- they are gathered and ranked against each other directly.
    __shared__ unsigned bins[kSelThreads];
    __shared__ unsigned wsum[kSelThreads / kWave];
    __shared__ unsigned s_digit, s_above, s_count, s_ncand;
    __shared__ unsigned long long cand[kSelThreads];
    int top = 63 - __clzll((long long)(kmin ^ kmax));  // keys agree above this bit
    unsigned long long prefix = top == 63 ? 0ull : (kmax >> (top + 1)) << (top + 1);
    unsigned remaining = remaining0;
    for (;;) {
        const int shift = top >= 9 ? top - 9 : 0, width = top - shift + 1;
        const unsigned dmask = (1u << width) - 1u;
        const unsigned long long himask = top == 63 ? 0ull : (~0ull << (top + 1));
        bins[tid] = 0u;
        __syncthreads();
        if (use_regs) {
#pragma unroll
            for (int k = 0; k < kSelPerThread; ++k) {
                const int64_t i = (int64_t)k * kSelThreads + tid;
                if ((int64_t)k * kSelThreads < Ptot)
                    hist_add(bins, (i < Ptot && (key[k] & himask) == prefix) ? (int)((unsigned)(key[k] >> shift) & dmask) : -1,
                             lane);
            }
        } else {
            // (uniform trip count: hist_add votes across the wave; a slot costs ~30 instructions of voting whether it
            //  holds a key or not -- 16 waves on 4 SIMDs, that, not the atomics, is the step's time -- so only the slots a
            //  few-thousand-key window fills are walked)
            for (int64_t i0 = tid - lane; i0 < Peff; i0 += (int64_t)kSelThreads * 8) {
                unsigned long long kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = i0 + lane + (int64_t)u * kSelThreads;
                    if (from_window)
                        kk[u] = i < Peff ? s_win[i] : 0ull;
                    else
                        kk[u] = i < Peff ? sort_key(fit_at(i)) : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = i0 + lane + (int64_t)u * kSelThreads;
                    if (i0 + (int64_t)u * kSelThreads >= Peff) break;  // (uniform over the wave: no slot of it holds a key)
                    hist_add(bins, (i < Peff && (kk[u] & himask) == prefix) ? (int)((unsigned)(kk[u] >> shift) & dmask) : -1, lane);
                }
            }
        }
        __syncthreads();
        SEL_TP(4);
        // thread t owns bin t: inclusive suffix sums over the bins >= t (wave scan, then the waves above)
        const unsigned c = bins[tid];
        unsigned suf = c;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_down(suf, off, kWave);
            if (lane + off < kWave) suf += o;
        }
        if (lane == 0) wsum[wv] = suf;  // the wave's total
        __syncthreads();
        for (int k = wv + 1; k < kSelThreads / kWave; ++k) suf += wsum[k];
        const unsigned above = suf - c;  // keys (matching the prefix) with a larger digit than t
        if (above < remaining && suf >= remaining) {  // exactly one thread
            s_digit = (unsigned)tid;
            s_above = above;
            s_count = c;
        }
        __syncthreads();
        prefix |= (unsigned long long)s_digit << shift;
        remaining -= s_above;
        const unsigned cnt = s_count;
        SEL_TP(5);
        if (shift == 0) {  // every bit fixed: the candidates are all equal to the prefix
            if (tid == 0) out[1] = prefix;
            return;
        }
        if (cnt <= 512u) {  // ranking costs ~11 ns per candidate, a further histogram step ~6.5 us
            const unsigned long long lomask = ~0ull << shift;
            if (tid == 0) s_ncand = 0u;
            __syncthreads();
            if (use_regs) {
#pragma unroll
                for (int k = 0; k < kSelPerThread; ++k) {
                    const int64_t i = (int64_t)k * kSelThreads + tid;
                    if ((int64_t)k * kSelThreads < Ptot && i < Ptot && (key[k] & lomask) == prefix)
                        cand[atomicAdd(&s_ncand, 1u)] = key[k];
                }
            } else {
                if (from_window) {
                    for (int i = tid; i < (int)Peff; i += kSelThreads) {
                        const unsigned long long kk = s_win[i];
                        if ((kk & lomask) == prefix) cand[atomicAdd(&s_ncand, 1u)] = kk;
                    }
                } else {
                    for (int64_t i = tid; i < Peff; i += kSelThreads) {
                        const unsigned long long kk = sort_key(fit_at(i));
                        if ((kk & lomask) == prefix) cand[atomicAdd(&s_ncand, 1u)] = kk;
                    }
                }
            }
            __syncthreads();
            SEL_TP(6);
            if ((unsigned)tid < cnt) {  // the remaining-th largest candidate: `greater` < remaining <= greater + equal
                const unsigned long long mine = cand[tid];
                unsigned greater = 0u, equal = 0u;
#pragma unroll 8
                for (unsigned cc = 0; cc < cnt; ++cc) {
                    const unsigned long long o = cand[cc];
                    greater += o > mine;
                    equal += o == mine;
                }
                if (greater < remaining && remaining <= greater + equal) out[1] = mine;  // same value from every tie
            }
            SEL_TP(7);
            return;
        }
        top = shift - 1;
    }
}

// One workgroup: radius = max(part_r)/sqrt(4n); if radius < delta, nw = int((P-1)/(1+exp((it/maxiter-gamma+0.5)/0.09)))
// and the nw-th largest pbestfit is found by a radix descent over keys held in registers, starting at the first bit in
// which the keys differ at all (see below).
// out[0] = nw (0 = no restart), out[1] = threshold key (rows with key >= threshold restart), out[2] = radius bits: the exact
// swarm radius here and in pso_restart_select_gathered; in cpso_post_kernel (the two-launch graph form) it is exact only when
// the decision needed the radius (kRadiusKnown / the exact pass) and otherwise the radius against the OLD best, within
// ||g_new - g_old|| / sqrt(4n) of it -- the restart decision is the same either way, the number is diagnostic
// The swarm is `nseg` segments (one per rank; 1 on a single GPU) of `seg_len` fitness values followed by
// `seg_npart` partial radii, `seg_stride` doubles apart: fit = base, part_r = base + seg_len.
__global__ __launch_bounds__(kSelThreads) void pso_restart_select_kernel(const sx_pso_args a,
                                                                         const double *__restrict__ fit,
                                                                         const double *__restrict__ part_r,
                                                                         int nseg, int64_t seg_len, int64_t seg_npart,
                                                                         int64_t seg_stride, double delta, double gamma,
                                                                         unsigned long long *__restrict__ out) {
    const int64_t Ptot = (int64_t)nseg * seg_len, npart = (int64_t)nseg * seg_npart;
    // element k of the (segmented) radius array; 32-bit arithmetic, nothing for one segment
    const unsigned unp = (unsigned)seg_npart;
    __shared__ double smax[kSelThreads / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    SEL_TP(0);
    const int done = a.state->done;
    const int64_t it = a.state->it;  // (fetched together, and looked at behind the radii: every dependent load is a trip to L2)
    double m = 0.0;
    for (int64_t k = tid; k < npart; k += kSelThreads) {
        const unsigned sg = nseg == 1 ? 0u : (unsigned)k / unp;
        m = fmax(m, part_r[(int64_t)sg * seg_stride + ((unsigned)k - sg * unp)]);
    }
    if (done) {
        if (tid == 0) out[0] = 0;
        return;
    }
    m = wave_max_f64(m);
    if (lane == 0) smax[wv] = m;
    __syncthreads();
    m = smax[0];
#pragma unroll
    for (int k = 1; k < kSelThreads / kWave; ++k) m = fmax(m, smax[k]);
    const double radius = m / sqrt(4.0 * (double)a.n);
    int64_t nw = 0;
    if (radius < delta) {
        const double inorm = (double)it / (double)a.maxiter;
        nw = (int64_t)(((double)Ptot - 1.0) / (1.0 + exp(1.0 / 0.09 * (inorm - gamma + 0.5))));
    }
    if (tid == 0) {
        out[0] = (unsigned long long)(nw > 0 ? nw : 0);
        out[2] = (unsigned long long)__double_as_longlong(radius);
    }
    SEL_TP(1);
    if (nw <= 0) return;  // uniform
    restart_threshold(a, fit, nseg, seg_len, seg_stride, Ptot, nw, out);
}

// ---------------------------------------------------------------------------
// CPSO inside a graph, whole-batch rows (n = 4 * LPR): ONE kernel behind the generation kernel -- best / termination
// (select_finalize_kernel's arithmetic on the in-place swarm), the restart question, and the selection when a restart is due.
// The generation kernel leaves r = max_i ||X_i - g_old||_2 per workgroup (its RAD by-product, in pso_radius_kernel's order of
// operations); with the step of the best d = ||g_new - g_old|| (= dx) radius_decision (sx_device.hpp) settles `radius < delta`
// in all but a handful of generations (C3b, 1 199 generations: best unchanged 288, above 583, below 324, undecided 4:
// tools/cpso_radius_decisions.py); the undecided ones get the radius from this workgroup's own pass over X.  Before: four
// dependent launches per generation (generation, best / termination, radius pass over X, selection).
// ---------------------------------------------------------------------------
constexpr int kPostHelpers = 63;             // helper workgroups of cpso_post_kernel's rare all-rows radius pass
constexpr long long kPostHelperWaitTicks = 20000;  // 200 us of the 100 MHz wall clock: how long a helper waits for workgroup 0's word
constexpr int kPostDxThreads = 256;  // the step of the best is summed exactly as select_finalize_kernel (256 threads) does
template <int LPR>
__global__ __launch_bounds__(kSelThreads) void cpso_post_kernel(const sx_pso_args a, const double *__restrict__ part_f,
                                                                const int64_t *__restrict__ part_i,
                                                                const double *__restrict__ part_rold, const int64_t npart,
                                                                const double xtol, const double delta, const double gamma,
                                                                unsigned long long *__restrict__ out, const int force_exact,
                                                                unsigned long long *__restrict__ hscratch) {
    constexpr int NW = kSelThreads / kWave, n = 4 * LPR;
    __shared__ double sf[NW];
    __shared__ int64_t si[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- helper workgroups (round 6; blockIdx >= 1, kPostHelpers of them) ------------------------------------------------
    // The rare exact-radius pass over ALL rows took ~570 us on this kernel's one workgroup (32 MB of X; the whole chip does it
    // in 7).  Now the launch has 1 + kPostHelpers workgroups: the helpers wait (bounded) for workgroup 0's word
    // hscratch[0] = {generation : 32 | help wanted : 1 | best row : 31} -- one agent-scope store, polled past the L1 -- and leave
    // at once when no help is wanted (every ordinary generation: they are gone before workgroup 0 is).  Wanted: helper h takes
    // slice h of the rows, stores its maximum (hscratch[2 + h]), drains, stores its tag (hscratch[2 + 64 + h] = generation);
    // workgroup 0 takes slice 0, waits (bounded) for the tags and, should a helper not have answered, does that slice itself --
    // the same per-row values and a maximum, which has no order.
    auto slice_radius = [&](int64_t r0, int64_t r1, const double *__restrict__ srcrow) -> double {
        constexpr int RPW = kWave / LPR, kRows = 8;
        const int l = lane & (LPR - 1), sub = lane / LPR;
        double gn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) gn[t] = srcrow[l + t * LPR];
        double mx = 0.0;
        for (int64_t j0 = r0 + (int64_t)wv * RPW + sub; j0 < r1; j0 += (int64_t)NW * RPW * kRows) {
            double xv[kRows][4];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                const int64_t j = j0 + (int64_t)q * NW * RPW, row = j < r1 ? j : r1 - 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[q][t] = a.X[row * a.ld + l + t * LPR];
            }
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                double ac = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double d = xv[q][t] - gn[t];
                    ac += d * d;
                }
                ac = sqrt(row_sum<LPR>(ac));
                if (j0 + (int64_t)q * NW * RPW < r1) mx = fmax(mx, ac);
            }
        }
        mx = wave_max_f64(mx);
        __syncthreads();
        if (lane == 0) sf[wv] = mx;
        __syncthreads();
        double m = sf[0];
        for (int k = 1; k < NW; ++k) m = fmax(m, sf[k]);
        return m;
    };
    const int64_t per_slice = (a.P + kPostHelpers) / (kPostHelpers + 1);
    if (blockIdx.x != 0) {
        if (hscratch == nullptr || a.state->done) return;
        const unsigned it32 = (unsigned)(a.state->it + 1);
        __shared__ unsigned long long s_word;
        if (tid == 0) {
            unsigned long long w = 0ull;
            const long long t0 = wall_clock64();
            for (;;) {
                w = __hip_atomic_load((__attribute__((address_space(1))) unsigned long long *)hscratch, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) == it32) break;
                if (wall_clock64() - t0 > kPostHelperWaitTicks) {
                    w = 0ull;  // (workgroup 0 never spoke: leave; it will do the work itself if it needed help)
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            s_word = w;
        }
        __syncthreads();
        const unsigned long long w = s_word;
        if (((w >> 31) & 1ull) == 0ull) return;
        const int64_t brow = (int64_t)(w & 0x7fffffffull);
        const int h = (int)blockIdx.x;
        const int64_t r0 = (int64_t)h * per_slice, r1 = r0 + per_slice < a.P ? r0 + per_slice : a.P;
        const double m = r0 < r1 ? slice_radius(r0, r1, a.pbest + brow * a.ld) : 0.0;
        if (tid == 0) {
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)(hscratch + 2 + h),
                               (unsigned long long)__double_as_longlong(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)(hscratch + 2 + 64 + h),
                               (unsigned long long)it32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // the records and the radii (npart <= 4 per thread at BASELINE config 3), requested before the state word is looked at
    double bf = __builtin_huge_val(), rold = 0.0;
    int64_t bi = INT64_MAX;
    for (int64_t k0 = tid; k0 < npart; k0 += 4 * kSelThreads) {
        double f[4], r[4];
        int64_t i[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = k0 + (int64_t)u * kSelThreads, kc = k < npart ? k : npart - 1;
            f[u] = part_f[kc], i[u] = part_i[kc], r[u] = part_rold[kc];
            if (k >= npart) f[u] = __builtin_huge_val(), i[u] = INT64_MAX;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            argmin_combine(bf, bi, f[u], i[u]);
            rold = fmax(rold, r[u]);
        }
    }
    const int done0 = a.state->done;
    const int64_t it = a.state->it + 1;  // the generation being finalised
    auto tell_helpers = [&](bool wanted, int64_t brow) {  // (thread 0; helpers see `done` themselves when the run is over)
        if (tid == 0 && hscratch != nullptr)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long *)hscratch,
                               ((unsigned long long)(unsigned)it << 32) | (wanted ? 0x80000000ull : 0ull) |
                                   (unsigned long long)(brow & 0x7fffffff),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (done0) {
        if (tid == 0) out[0] = 0;
        return;
    }
    wave_argmin_all(bf, bi);
    rold = wave_max_f64(rold);
    if (lane == 0) sf[wv] = bf, si[wv] = bi;
    __syncthreads();
    bf = sf[0], bi = si[0];
    for (int k = 1; k < NW; ++k) argmin_combine(bf, bi, sf[k], si[k]);
    __syncthreads();
    if (lane == 0) sf[wv] = rold;
    __syncthreads();
    double r = sf[0];
    for (int k = 1; k < NW; ++k) r = fmax(r, sf[k]);
    r = sqrt(r);  // (the generation kernel leaves squared radii)
    __syncthreads();
    // dx = ||g_old - pbest[best]||_2, thread t < 256 the elements t, t + 256, ... (n <= 256: one each), then the wave, then the waves
    const double *__restrict__ src = a.pbest + bi * a.ld;
    double gv = 0.0, sv = 0.0, acc = 0.0;
    const bool mine = tid < kPostDxThreads && tid < n;
    if (mine) gv = a.gbest[tid], sv = src[tid];
    if (mine) {
        const double d = gv - sv;
        acc += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) sf[wv] = acc;
    __syncthreads();
    double ss = 0.0;
    for (int k = 0; k < kPostDxThreads / kWave; ++k) ss += sf[k];
    const double dx = sqrt(ss);
    if (mine) a.gbest[tid] = sv;
    int status = SX_STATUS_NONE;
    if (dx <= xtol && bf <= a.ftol)
        status = 0;
    else if (bf <= a.ftol)
        status = 1;
    else if (it >= a.maxiter)
        status = -1;
    if (tid == 0) {
        sx_state *st = a.state;
        st->it = it;
        st->gbidx = bi;
        st->gfit = bf;
        st->dx = dx;
        st->status = status;
        st->done = status != SX_STATUS_NONE;
    }
    if (status != SX_STATUS_NONE) {  // (what the selection does when it finds the run over)
        if (tid == 0) out[0] = 0;
        tell_helpers(false, 0);
        return;
    }
    // ---- the restart question, cpso/_cpso.py:405-412 ----
    // (force_exact: SX_CPSO_FORCE_EXACT=1 / 2 at graph creation -- every generation takes the rare branch / its all-rows form; tests)
    const unsigned long long dec = force_exact ? kRadiusExactNeeded : radius_decision(r, dx, delta, n);
    double m = r;
    if (dec != kRadiusExactNeeded) tell_helpers(false, 0);
    if (dec == kRadiusExactNeeded) {
        // Rare (4 of 1 199 generations at C3b): the radius against the new best (= pbest[best], what gbest holds from now on),
        // row by row as pso_radius_kernel does it -- but only for the rows that can hold the maximum.  A row whose radius
        // against the OLD best (left per row by the generation kernel, behind the per-workgroup maxima) is below
        // r - 2 d - 1e-6 r ends below r - d - 1e-6 r, and the row that attains r ends at r - d or above: the maximum over the
        // candidates is the maximum over all rows, the same floating-point values.  More candidates than the list holds (the
        // first generations, when the best still jumps): all rows, 8 per wavefront in flight (~640 us at P = 16384).
        constexpr int RPW = kWave / LPR, kCand = 4096;
        __shared__ int s_cand[kCand];
        __shared__ unsigned s_ncand;
        const double *__restrict__ row_rold = part_rold + npart;
        if (tid == 0) s_ncand = 0u;
        __syncthreads();
        const double cut1 = (r - 2.0 * dx) - 1.0e-6 * r;
        const double cut = cut1 > 0.0 ? cut1 * cut1 * (1.0 - 1.0e-12) : -1.0;  // (against the SQUARED per-row radii)
        if (force_exact != 2) {
            for (int64_t i0 = tid; i0 < a.P; i0 += 8 * kSelThreads) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = i0 + (int64_t)u * kSelThreads;
                    v[u] = row_rold[i < a.P ? i : a.P - 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = i0 + (int64_t)u * kSelThreads;
                    if (i < a.P && v[u] >= cut) {
                        const unsigned k = atomicAdd(&s_ncand, 1u);
                        if (k < (unsigned)kCand) s_cand[k] = (int)i;
                    }
                }
            }
        }
        __syncthreads();
        const unsigned nc = s_ncand;
        // (a list is worth walking alone while it is short: 64 workgroups take ALL rows in the time this one takes 256)
        const bool helpers = hscratch != nullptr && gridDim.x == (unsigned)(kPostHelpers + 1);
        const bool listed = force_exact != 2 && nc <= (unsigned)(helpers ? 256 : kCand);  // (uniform)
        tell_helpers(!listed && helpers, bi);
        if (!listed && helpers) {
            double mall = slice_radius(0, per_slice < a.P ? per_slice : a.P, src);
            __shared__ unsigned long long s_tag[kPostHelpers + 1];
            __shared__ int s_all;
            const long long t0 = wall_clock64();
            for (;;) {  // wave 0: the helpers' tags
                if (tid < kWave) {
                    const int h = tid;
                    unsigned long long tg = (unsigned)it;
                    if (h >= 1 && h <= kPostHelpers)
                        tg = __hip_atomic_load((__attribute__((address_space(1))) unsigned long long *)(hscratch + 2 + 64 + h),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_tag[h] = tg;
                    const bool ok = __all(tg == (unsigned long long)(unsigned)it);
                    if (tid == 0) s_all = ok ? 1 : (wall_clock64() - t0 > 4 * kPostHelperWaitTicks ? -1 : 0);
                }
                __syncthreads();
                const int st_all = s_all;
                __syncthreads();
                if (st_all != 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
            for (int h = 1; h <= kPostHelpers; ++h) {  // (uniform: s_tag is shared)
                const int64_t r0 = (int64_t)h * per_slice, r1 = r0 + per_slice < a.P ? r0 + per_slice : a.P;
                double mh = 0.0;
                if (s_tag[h] == (unsigned long long)(unsigned)it)
                    mh = __longlong_as_double((long long)__hip_atomic_load(
                        (__attribute__((address_space(1))) unsigned long long *)(hscratch + 2 + h), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT));
                else if (r0 < r1)
                    mh = slice_radius(r0, r1, src);  // (a helper that never answered: its slice here)
                mall = fmax(mall, mh);
            }
            m = mall;
        } else {
        const int64_t total = listed ? (int64_t)nc : a.P;
        const int l = lane & (LPR - 1), sub = lane / LPR;
        double gn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) gn[t] = src[l + t * LPR];
        double mx = 0.0;
        constexpr int kRows = 8;
        for (int64_t j0 = (int64_t)wv * RPW + sub; j0 < total; j0 += (int64_t)NW * RPW * kRows) {
            double xv[kRows][4];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                const int64_t j = j0 + (int64_t)q * NW * RPW, jc = j < total ? j : total - 1;
                const int64_t row = listed ? (int64_t)s_cand[jc] : jc;
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[q][t] = a.X[row * a.ld + l + t * LPR];
            }
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                double ac = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double d = xv[q][t] - gn[t];
                    ac += d * d;
                }
                ac = sqrt(row_sum<LPR>(ac));
                if (j0 + (int64_t)q * NW * RPW < total) mx = fmax(mx, ac);
            }
        }
        mx = wave_max_f64(mx);
        if (lane == 0) sf[wv] = mx;
        __syncthreads();
        m = sf[0];
        for (int k = 1; k < NW; ++k) m = fmax(m, sf[k]);
        }
    }
    const double radius = m / sqrt(4.0 * (double)n);  // (kRadiusAbove / kRadiusBelow: of r, within d of the swarm's)
    int64_t nw = 0;
    if (dec == kRadiusBelow || (dec != kRadiusAbove && radius < delta)) {
        const double inorm = (double)it / (double)a.maxiter;
        nw = (int64_t)(((double)a.P - 1.0) / (1.0 + exp(1.0 / 0.09 * (inorm - gamma + 0.5))));
    }
    if (tid == 0) {
        out[0] = (unsigned long long)(nw > 0 ? nw : 0);
        out[2] = (unsigned long long)__double_as_longlong(radius);
    }
    if (nw <= 0) return;  // uniform
    restart_threshold(a, a.pbestfit, 1, a.P, a.P, a.P, nw, out);
}

// rows whose pbestfit is among the nw worst: V = 0, X = uniform(lower, upper), pbest = X, pbestfit = 1e30 (:420-424)
// host_rows != NULL (numpy-legacy): row ids in the reference's descending-fitness order + their new positions
__global__ __launch_bounds__(256) void pso_restart_apply_kernel(
    const sx_pso_args a, const unsigned long long *__restrict__ sel, const int64_t *__restrict__ host_rows,
    const double *__restrict__ host_x, int64_t host_count) {
    if (a.state->done) return;
    const int lane = (int)(threadIdx.x & 63);
    const int64_t slot = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t row;
    if (host_rows != nullptr) {
        if (slot >= host_count) return;
        row = host_rows[slot];
    } else {
        if (slot >= a.P || sel[0] == 0) return;
        row = slot;
        if (sort_key(a.pbestfit[row]) < sel[1]) return;
    }
    const uint32_t gen = (uint32_t)a.state->it;  // the generation that just finished
    const uint32_t grow = (uint32_t)(a.row0 + row);
    double *__restrict__ xr = a.X + row * a.ld;
    double *__restrict__ vr = a.V + row * a.ld;
    double *__restrict__ pb = a.pbest + row * a.ld;
    for (int e = lane; e < a.n; e += kWave) {
        double x;
        if (host_rows != nullptr)
            x = host_x[slot * (int64_t)a.n + e];
        else
            x = a.lower[e] + (a.upper[e] - a.lower[e]) *
                                 philox_u53(e, lanes_per_row(a.n), grow, gen, kPurposePsoRestart, a.key0, a.key1);
        vr[e] = 0.0;
        xr[e] = x;
        pb[e] = x;
    }
    if (lane == 0) a.pbestfit[row] = 1.0e30;
}

}  // namespace

extern "C" int sx_pso_generation(const sx_pso_args *a, int finalize, void *stream) {
    if (int rc = check_args(a)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const Geometry g = geometry(a->P, a->n);
    if (is_wide(a->n)) {  // rows of more than sx_wide_from() elements: one workgroup per particle (sx_wide.hip)
        if (int rc = wide_pso_launch(a, s)) return rc;
    } else {
        PlanArg plan;
        if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
        const bool plain = a->constraints == 0 && a->pending_restart == nullptr;
        pso_kernel_t kern = a->rng == SX_RNG_PHILOX ? pick_kernel<SX_RNG_PHILOX>(a->fun_id, a->n, plain)
                                                     : pick_kernel<SX_RNG_HOST>(a->fun_id, a->n, plain);
        hipLaunchKernelGGL(kern, dim3(g.blocks), dim3(g.threads), g.lds, s, *a, plan, (double *)nullptr, 0, 0, (int64_t)0);
        SX_LAUNCH_CHECK();
    }
    if (finalize)
        return sx_select_finalize(a->part_f, a->part_i, g.blocks, a->pbest, a->pbest, a->ld, a->n, a->gbest, a->state,
                                  a->maxiter, a->xtol, a->ftol, stream);
    return 0;
}

extern "C" int sx_pso_radius(const sx_pso_args *a, double *part_r, void *stream) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(part_r != nullptr, "sx_pso_radius: null scratch");
    const Geometry g = geometry(a->P, a->n);
    SX_DISPATCH_LPR(a->n, hipLaunchKernelGGL(pso_radius_kernel<LPR>, dim3(g.blocks), dim3(g.threads), 0,
                                             (hipStream_t)stream, *a, part_r))
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_pso_restart_select(const sx_pso_args *a, const double *part_r, double delta, double gamma,
                                     uint64_t *out3, void *stream) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(part_r && out3, "sx_pso_restart_select: null pointer");
    const Geometry g = geometry(a->P, a->n);
    hipLaunchKernelGGL(pso_restart_select_kernel, dim3(1), dim3(kSelThreads), 0, (hipStream_t)stream, *a,
                       (const double *)a->pbestfit, part_r, 1, a->P, (int64_t)g.blocks, a->P, delta, gamma,
                       (unsigned long long *)out3);
    SX_LAUNCH_CHECK();
    return 0;
}

// Sharded swarm: `gathered` = (world, P_local + npart) doubles, row r = rank r's [pbestfit | partial radii]
// (one all-gather per generation); every rank derives the same radius / nw / threshold for the WHOLE swarm.
extern "C" int sx_pso_restart_select_gathered(const sx_pso_args *a, const double *gathered, int world, double delta,
                                              double gamma, uint64_t *out3, void *stream) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(gathered && out3 && world >= 1, "sx_pso_restart_select_gathered: bad arguments");
    const int64_t npart = (int64_t)geometry(a->P, a->n).blocks;
    hipLaunchKernelGGL(pso_restart_select_kernel, dim3(1), dim3(kSelThreads), 0, (hipStream_t)stream, *a, gathered,
                       gathered + a->P, world, a->P, npart, a->P + npart, delta, gamma, (unsigned long long *)out3);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_pso_restart_apply(const sx_pso_args *a, const uint64_t *sel3, const int64_t *host_rows,
                                    const double *host_x, int64_t host_count, void *stream) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE((sel3 != nullptr) != (host_rows != nullptr), "sx_pso_restart_apply: give sel3 OR host rows");
    SX_REQUIRE(host_rows == nullptr || (host_x != nullptr && host_count >= 0), "sx_pso_restart_apply: host rows");
    SX_REQUIRE(a->lower && a->upper, "sx_pso_restart_apply: bounds missing");
    const int64_t slots = host_rows ? host_count : a->P;
    if (slots == 0) return 0;
    const int rpb = 4;
    hipLaunchKernelGGL(pso_restart_apply_kernel, dim3((unsigned)((slots + rpb - 1) / rpb)), dim3(rpb * kWave), 0,
                       (hipStream_t)stream, *a, (const unsigned long long *)sel3, host_rows, host_x, host_count);
    SX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// hipGraph of `ngen` generations (single GPU, Philox draws): per generation the generation kernel, the
// best/termination kernel and -- CPSO (part_r != NULL) -- the three restart kernels, all reading the
// generation counter and the done flag from the device, so one instantiated graph is replayed.
// ---------------------------------------------------------------------------
namespace {
int add_kernel_node(hipGraph_t graph, hipGraphNode_t *prev, void *func, dim3 grid, dim3 block, unsigned lds,
                    void **kargs) {
    hipKernelNodeParams kp = {};
    kp.func = func;
    kp.gridDim = grid;
    kp.blockDim = block;
    kp.sharedMemBytes = lds;
    kp.kernelParams = kargs;
    kp.extra = nullptr;
    hipGraphNode_t node;
    SX_HIP(hipGraphAddKernelNode(&node, graph, *prev ? prev : nullptr, *prev ? 1 : 0, &kp));
    *prev = node;
    return 0;
}

bool fused_radius_off() {  // (read when a graph is created, so that one process can build both forms)
    const char *e = getenv("SX_CPSO_FUSED_RADIUS");
    return e != nullptr && e[0] == '0';
}
template <int LPR>
void *radius_kernel_ptr() {
    return (void *)pso_radius_kernel<LPR>;
}
template <int LPR>
void *post_kernel_ptr() {
    return (void *)cpso_post_kernel<LPR>;
}
}  // namespace

namespace sx {
int add_finalize_node(hipGraph_t graph, hipGraphNode_t *prev, const double *part_f, const int64_t *part_i,
                      int64_t npart, const double *rows0, const double *rows1, int64_t ld, int n, double *gbest,
                      sx_state *state, int maxiter, double xtol, double ftol);
}

extern "C" int sx_pso_graph_create(const sx_pso_args *a, int ngen, double *part_r, double delta, double gamma,
                                   uint64_t *sel3, sx_graph **out) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(out != nullptr && ngen >= 1, "sx_pso_graph_create: bad arguments");
    SX_REQUIRE(a->rng == SX_RNG_PHILOX, "sx_pso_graph_create: graphs need in-kernel (Philox) draws");
    SX_REQUIRE((part_r == nullptr) == (sel3 == nullptr), "sx_pso_graph_create: restart needs part_r AND sel3");
    SX_REQUIRE(part_r == nullptr || (a->lower && a->upper), "sx_pso_graph_create: bounds missing");
    PlanArg plan = {};
    const bool wide = is_wide(a->n);
    if (!wide && make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a->P, a->n);
    sx_graph *gr = new sx_graph();
    SX_HIP(hipGraphCreate(&gr->graph, 0));
    // CPSO with whole-batch rows: two launches per generation (generation kernel with its radius by-product, cpso_post_kernel)
    // instead of four -- C3b 51.7 -> 46.9 us per generation (profiles/r4_cpso_fused_radius.txt).  The graph owns the scratch (npart partial radii).
    // SX_CPSO_FUSED_RADIUS=0: best / termination, the radius pass over X and the selection as launches of their own.
    const bool fused_radius = part_r != nullptr && a->n == 4 * lanes_per_row(a->n) && a->n <= kPostDxThreads &&
                              !fused_radius_off();
    double *part_rold = nullptr;
    if (fused_radius) {
        // (npart per-workgroup maxima, then P per-row radii against the old best: what the rare exact branch looks at first)
        // ... and 2 + 64 + 64 words for the post kernel's helper workgroups (their word, their maxima, their tags)
        SX_HIP(hipMalloc(&gr->scratch, ((size_t)g.blocks + (size_t)a->P + 130) * sizeof(double)));
        SX_HIP(hipMemset(gr->scratch, 0, ((size_t)g.blocks + (size_t)a->P + 130) * sizeof(double)));
        part_rold = (double *)gr->scratch;
    }
    sx_pso_args args = *a;
    args.pending_restart = nullptr;
    // generations 2..ngen of the graph carry out the previous generation's restart themselves (sx_pso_args.pending_restart);
    // the last one is followed by the apply kernel, so the state a replay leaves behind is complete
    sx_pso_args args_inline = args;
    args_inline.pending_restart = sel3;
    double *no_rows_buf = nullptr;
    int izero = 0;
    int64_t npart_gen = fused_radius ? (int64_t)g.blocks : 0;
    void *gen_args[] = {&args, &plan, fused_radius ? &part_rold : &no_rows_buf, &izero, &izero, &npart_gen};
    void *gen_args_inline[] = {&args_inline, &plan, fused_radius ? &part_rold : &no_rows_buf, &izero, &izero, &npart_gen};
    // restart kernels' arguments
    const double *fit = a->pbestfit;
    const double *pr = part_r;
    int one = 1;
    int64_t P = a->P, npart = g.blocks;
    unsigned long long *sel = (unsigned long long *)sel3;
    void *rad_args[] = {&args, &part_r};
    void *sel_args[] = {&args, &fit, &pr, &one, &P, &npart, &P, &delta, &gamma, &sel};
    const double *cpart_f = a->part_f, *cpart_rold = part_rold;
    const int64_t *cpart_i = a->part_i;
    double xtol = a->xtol;
    // (tests: 1 = every generation through the exact branch, 2 = through its all-rows form)
    const char *fe = getenv("SX_CPSO_FORCE_EXACT");
    int force_exact = fe == nullptr ? 0 : (fe[0] == '2' ? 2 : 1);
    // (SX_CPSO_HELPERS=0: the one-workgroup launch of rounds 4-5)
    const char *he = getenv("SX_CPSO_HELPERS");
    const bool post_helpers = fused_radius && !(he != nullptr && he[0] == '0') && a->P < (int64_t)0x7fffffff;
    unsigned long long *hscratch = post_helpers ? (unsigned long long *)((double *)gr->scratch + (size_t)g.blocks + (size_t)a->P) : nullptr;
    void *post_args[] = {&args, &cpart_f, &cpart_i, &cpart_rold, &npart, &xtol, &delta, &gamma, &sel, &force_exact, &hscratch};
    void *post_fn = nullptr;
    SX_DISPATCH_LPR(a->n, post_fn = post_kernel_ptr<LPR>())
    const int64_t *no_rows = nullptr;
    const double *no_x = nullptr;
    int64_t zero = 0;
    const unsigned long long *csel = sel;
    void *app_args[] = {&args, &csel, &no_rows, &no_x, &zero};
    void *radius_fn = nullptr;
    SX_DISPATCH_LPR(a->n, radius_fn = radius_kernel_ptr<LPR>())
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < ngen; ++i) {
        const bool inl = part_r != nullptr && i > 0;  // CPSO: generations 2.. carry out the restart decided before them
        // (the first generation of a fused-radius graph takes the general kernel too: the plain one leaves no radius)
        if (wide) {
            if (int rc = wide_pso_add_node(gr->graph, &prev, inl ? &args_inline : &args)) return rc;
        } else if (int rc = add_kernel_node(gr->graph, &prev,
                                            (void *)pick_kernel<SX_RNG_PHILOX>(a->fun_id, a->n,
                                                                               a->constraints == 0 && !inl && !fused_radius),
                                            dim3(g.blocks), dim3(g.threads), (unsigned)g.lds,
                                            inl ? gen_args_inline : gen_args))
            return rc;
        if (fused_radius) {
            if (int rc = add_kernel_node(gr->graph, &prev, post_fn, dim3(post_helpers ? kPostHelpers + 1 : 1), dim3(kSelThreads), 0,
                                         post_args))
                return rc;
        } else if (int rc = add_finalize_node(gr->graph, &prev, a->part_f, a->part_i, g.blocks, a->pbest, a->pbest, a->ld,
                                              a->n, a->gbest, a->state, a->maxiter, a->xtol, a->ftol))
            return rc;
        if (part_r != nullptr) {
            if (!fused_radius) {
                if (int rc = add_kernel_node(gr->graph, &prev, radius_fn, dim3(g.blocks), dim3(g.threads), 0, rad_args))
                    return rc;
                if (int rc = add_kernel_node(gr->graph, &prev, (void *)pso_restart_select_kernel, dim3(1),
                                             dim3(kSelThreads), 0, sel_args))
                    return rc;
            }
            const int rpb = 4;
            if (i == ngen - 1)
                if (int rc = add_kernel_node(gr->graph, &prev, (void *)pso_restart_apply_kernel,
                                             dim3((unsigned)((a->P + rpb - 1) / rpb)), dim3(rpb * kWave), 0, app_args))
                    return rc;
        }
    }
    SX_HIP(hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0));
    *out = gr;
    return 0;
}

// ---------------------------------------------------------------------------
// PSO, one kernel per generation ("chained": the best / termination step of generation g runs in the prologue of
// the launch that produces g + 1; see pso_generation_kernel<..., CHAIN>).  Single GPU, in-kernel draws,
// constraints=None, whole-batch rows (n = 64, 128, 256), no competitive restart.
//   a->state   3 sx_state words: [0], [1] ping-pong, [2] the host's view (finalize_only launches)
//   a->part_f / part_i   2 x npart records (rec = 2 * row + q)
//   best_rows  2 x npart x n doubles: every workgroup's copy of its best row, double-buffered by q
// ---------------------------------------------------------------------------
extern "C" int sx_pso_chain_supported(const sx_pso_args *a) {
    if (check_args(a)) return 0;
    const Geometry g = geometry(a->P, a->n);
    return hot_objective(a->fun_id) && a->rng == SX_RNG_PHILOX && a->constraints == 0 && a->pending_restart == nullptr &&
                   a->n == 4 * lanes_per_row(a->n) && (int64_t)g.blocks <= (int64_t)kChainRecPerThread * g.threads
               ? 1
               : 0;
}

extern "C" int sx_pso_chain_launch(const sx_pso_args *a, double *best_rows, int parity, int finalize_only, void *stream) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(best_rows != nullptr && (parity == 0 || parity == 1), "sx_pso_chain_launch: bad arguments");
    SX_REQUIRE(sx_pso_chain_supported(a), "sx_pso_chain_launch: shape / mode not supported by the chained kernel");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a->P, a->n);
    hipLaunchKernelGGL(pick_chain_kernel(a->fun_id, a->n), dim3(finalize_only ? 1u : g.blocks), dim3(g.threads), g.lds,
                       (hipStream_t)stream, *a, plan, best_rows, parity, finalize_only ? 1 : 0, (int64_t)g.blocks);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_pso_chain_graph_create(const sx_pso_args *a, double *best_rows, int ngen, int start_parity,
                                         sx_graph **out) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(out != nullptr && ngen >= 1 && best_rows != nullptr && (start_parity == 0 || start_parity == 1),
               "sx_pso_chain_graph_create: bad arguments");
    SX_REQUIRE(sx_pso_chain_supported(a), "sx_pso_chain_graph_create: shape / mode not supported by the chained kernel");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a->P, a->n);
    sx_graph *gr = new sx_graph();
    SX_HIP(hipGraphCreate(&gr->graph, 0));
    sx_pso_args args = *a;
    int par[2] = {0, 1}, mode = 0;
    int64_t npart = g.blocks;
    void *kargs[2][6] = {{&args, &plan, &best_rows, &par[0], &mode, &npart}, {&args, &plan, &best_rows, &par[1], &mode, &npart}};
    void *fn = (void *)pick_chain_kernel(a->fun_id, a->n);
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < ngen; ++i)
        if (int rc = add_kernel_node(gr->graph, &prev, fn, dim3(g.blocks), dim3(g.threads), (unsigned)g.lds,
                                     kargs[(start_parity + i) & 1]))
            return rc;
    SX_HIP(hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0));
    *out = gr;
    return 0;
}
