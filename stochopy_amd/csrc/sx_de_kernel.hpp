// The DE generation kernel (all exchange modes) and its dispatch: included by one translation unit per exchange mode
// (sx_de.hip: XM = 0, sx_de_chain.hip: XM = 1, sx_de_p2p.hip: XM = 2 -- SX_DE_XM says which), so that the three
// families of instantiations compile in parallel.  Everything here has internal linkage.
#pragma once
#ifndef SX_DE_XM
#error "define SX_DE_XM (0, 1 or 2) before including sx_de_kernel.hpp"
#endif
#include <type_traits>
#include <vector>

#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"
#include "sx_xchg.hpp"

namespace sx {
int make_plan_arg(int fun_id, int n, PlanArg *out);
int add_finalize_node(hipGraph_t graph, hipGraphNode_t *prev, const double *part_f, const int64_t *part_i,
                      int64_t npart, const double *rows0, const double *rows1, int64_t ld, int n, double *gbest,
                      sx_state *state, int maxiter, double xtol, double ftol);
int check_xchg_args(const sx_xchg_args *x);
}
using namespace sx;

#if defined(SX_TRACE) && SX_DE_XM == 1
// debug build only: per-workgroup checkpoints (s_memrealtime, 100 MHz) of the (chained) generation kernel
__device__ unsigned long long sx_trace_buf[1024 * 16];
#define SX_TP(k)                                                                              \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                          \
            sx_trace_buf[blockIdx.x * 16 + (k)] = wall_clock64();                             \
            sx_trace_buf[blockIdx.x * 16 + 8 + (k)] = clock64();                              \
        }                                                                                     \
    } while (0)
extern "C" int sx_trace_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sx_trace_buf), sizeof(unsigned long long) * 1024 * 16);
}
#else
#define SX_TP(k) do {} while (0)
#endif

namespace {

// XM = 0: a.state is ONE sx_state, the best/termination step is a separate kernel.
// XM = 1 ("chained finalize", single GPU + Philox): a.state is sx_state[3], a.part_f/part_i are
// [2][npart].  Launch L (parity p = L & 1) first finalises the generation its predecessor produced --
// EVERY wavefront reduces the npart (<= 512) records part[p] (written before the kernel boundary, so
// plainly visible) and derives the same best / status; workgroup 0 publishes it in state[1-p] -- and then
// produces the next generation, writing records to part[1-p].  No second kernel, no atomics.
// mode 1 = finalise only (one workgroup), result to state[2] for the host.  dx (xtol) only separates
// status 0 from 1, both of which stop: the host derives it from state.reserved[0] (previous best row).
// XM = 2 (multi-GPU, peer exchange; sx_xchg.hpp): as XM = 1, but the population is this rank's shard.
// Workgroup 0 is a service workgroup: it reduces the shard's records, writes [f, global row, row] with
// the generation tag into every peer's exchange buffer (one wavefront per peer) and publishes the state;
// the row workgroups (blockIdx 1..) wait for the tagged headers of all ranks in their OWN rank's buffer,
// pick the global best and read its row from there.  Still one kernel per generation; the only
// cross-GPU dependency is "all ranks have reached this generation".  With x.global_rows > 0 the donor rows
// are drawn over the whole population and read from their owners' (IPC-mapped) population buffers.

// XM = 2, workgroup 0: shard best -> peers, global best -> state (and, for whole-wave rows, the winning record
// into the cacheable relay).  Sets *x.error on timeout.
__device__ __forceinline__ void p2p_service(const sx_de_args &a, const sx_xchg_args &x, int chain_p, int mode,
                                            int64_t npart, const sx_state *sin, bool relay) {
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), nw = (int)(blockDim.x >> 6);
    // the state word and the records do not depend on each other: fetch the state first (relaxed atomics pin
    // the loads here), scan the records, and only then look at it
    const int done0 = __atomic_load_n(&sin->done, __ATOMIC_RELAXED);
    const int64_t it0 = __atomic_load_n(&sin->it, __ATOMIC_RELAXED);
    const int64_t prev_winner = __atomic_load_n(&sin->reserved[1], __ATOMIC_RELAXED);
    const int err0 = __atomic_load_n(x.error, __ATOMIC_RELAXED);
    const double *pf = a.part_f + (int64_t)chain_p * npart;
    const int64_t *pi = a.part_i + (int64_t)chain_p * npart;
    // shard best = lexicographic (f, row) minimum of the records: the whole workgroup scans them (8 records per
    // thread and trip, loads overlapping), waves meet in LDS; every wave ends up with the same pair
    __shared__ double svc_f[kMaxWavesPerBlock];
    __shared__ int64_t svc_i[kMaxWavesPerBlock];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    if (npart <= 8 * kWave) {
        // few records (the latency-critical small shards): every wave scans all of them on its own, lane-
        // contiguous slices + the DPP minimum -- no LDS, no barrier (as the single-GPU chained kernel does)
        const int per = (int)((npart + kWave - 1) / kWave);
        const int64_t k0 = (int64_t)lane * per;
        double f[8];
        int64_t i[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = u < per && k0 + u < npart;
            f[u] = in ? pf[k0 + u] : __builtin_huge_val();
            i[u] = in ? pi[k0 + u] : INT64_MAX;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) argmin_combine(bf, bi, f[u], i[u]);
        wave_argmin_ordered(bf, bi);
    } else {
        for (int64_t k0 = threadIdx.x; k0 < npart; k0 += (int64_t)blockDim.x * 8) {
            double f[8];
            int64_t i[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t k = k0 + (int64_t)u * blockDim.x;
                f[u] = k < npart ? pf[k] : __builtin_huge_val();
                i[u] = k < npart ? pi[k] : INT64_MAX;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) argmin_combine(bf, bi, f[u], i[u]);
        }
        wave_argmin_all(bf, bi);
        if (lane == 0) {
            svc_f[wave] = bf;
            svc_i[wave] = bi;
        }
        __syncthreads();
        for (int w = 0; w < nw; ++w) argmin_combine(bf, bi, svc_f[w], svc_i[w]);
    }
    if (done0) {
        if (mode == 1 && threadIdx.x == 0) a.state[2] = *sin;
        return;
    }
    // an earlier wait timed out: the run is dead, the host raises.  With the relay a barrier follows: the whole
    // workgroup has to agree (the flag may have been raised between two wavefronts' loads)
    if (relay ? __syncthreads_or(err0) : err0) return;
    const int64_t it = it0 + 1;  // the generation the population holds
    const uint32_t tag = (uint32_t)(it + 1);
    const double *row = ((it & 1) ? a.buf1 : a.buf0) + bi * a.ld;
    for (int r = wave; r < x.world; r += nw)
        xchg_push_record(x.peer[r] + xchg_slot_offset(a.n, chain_p, x.rank), bf, a.row0 + bi, row, a.n, tag, lane);
    if (wave != 0 && !relay) return;
    double gf;
    int64_t gi;
    int winner;
    if (!xchg_wait_best(x.peer[x.rank] + xchg_slot_offset(a.n, chain_p, 0), a.n, x.world, tag, x.timeout_ticks, lane,
                        gf, gi, winner)) {
        if (lane == 0) atomicExch(x.error, 1);
        return;
    }
    if (relay) {
        // long rows: every row wavefront reading the winner's 16(n+2) bytes from uncached memory would cost
        // more than the generation's own traffic, so this workgroup copies the record once into ordinary
        // (L2-cacheable) memory, still as tagged words; device-scope stores, the readers verify the tags
        const uint64_t *src = x.peer[x.rank] + xchg_slot_offset(a.n, chain_p, winner);
        uint64_t *dst = x.relay + (int64_t)chain_p * xchg_relay_stride(a.n);
        const uint64_t t0 = wall_clock64();
        bool lost = false;
        const int nwords = 2 * (a.n + 2);
        for (int j0 = (int)threadIdx.x; j0 < nwords && !lost; j0 += (int)blockDim.x * 8) {
            uint64_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // 8 words per thread in flight; a late word is re-read on its own
                const int j = j0 + u * (int)blockDim.x;
                w[u] = j < nwords ? ll_load(src + j) : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * (int)blockDim.x;
                if (j >= nwords) continue;
                while (!ll_ok(w[u], tag)) {
                    if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
                        atomicExch(x.error, 1);
                        lost = true;
                        break;
                    }
                    w[u] = ll_load(src + j);
                }
                __hip_atomic_store(dst + j, w[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // all words written through, then the "ready" word: readers poll it past the caches and only then
        // touch the row with cacheable loads, so no cache ever holds a line from before the copy
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(dst + xchg_slot_words(a.n), (uint64_t)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (wave != 0) return;
    }
    if (lane == 0) {
        int status = SX_STATUS_NONE;
        if (it >= 2) {
            if (gf <= a.ftol)
                status = 1;
            else if (it >= a.maxiter)
                status = -1;
        }
        sx_state *so = a.state + (mode == 1 ? 2 : 1 - chain_p);
        so->it = it;
        so->gbidx = gi;  // GLOBAL row
        so->gfit = gf;
        so->dx = 0.0;
        so->status = status;
        so->done = status != SX_STATUS_NONE;
        so->reserved[0] = prev_winner;        // rank that held the previous best row
        so->reserved[1] = winner;
    }
}

constexpr int kStep = 4;  // row steps per batch: one Philox call, and all its loads in flight together

// FULL: n is a whole number of batches (n % (kStep*LPR) == 0) and P a whole number of workgroups, so
// every bounds test folds away (the P = 4096, n = 128 headline shape).
// NFIX: FULL with n == kStep * LPR exactly (64, 128 -- the headline shape -- or 256): the row length, and with it numpy's
// summation plan, is a compile-time constant (row_reduce_fixed / row_reduce_static in sx_device.hpp); 0 otherwise.
// STRAT: with NFIX and constraints=None, the strategy as a compile-time constant too (its donor count, the best-row
// fetch and the mutant's formula are otherwise uniform branches inside the row's dependent chain: 9.03 -> 8.52 us per
// generation at the headline shape), and no repair code; -1 = strategy and constraints read from the arguments.
// SX_DE_NFIX_WAVES (A/B): minimum wavefronts per SIMD asked of the one-batch kernels -- 6 = three 512-thread workgroups per CU,
// which caps them at 80 VGPRs (they take 84 uncapped: two workgroups per CU)
#ifndef SX_DE_NFIX_WAVES
#define SX_DE_NFIX_WAVES 1
#endif
// SX_DE_WAVE_ROWS_WAVES (A/B): the same for the whole-wave general kernels of the two-kernel path (88 / 107 VGPRs uncapped)
#ifndef SX_DE_WAVE_ROWS_WAVES
#define SX_DE_WAVE_ROWS_WAVES 1
#endif
// ONEB (round 5): a whole-wave kernel instantiated for rows of 129 ... 256 elements of run-time length -- one batch, rows in
// registers for the store, the objective's run-time register chain only (as the short-row kernels are for n <= 128)
template <int FUN, int RNG, int XM, int LPR, bool FULL, int NFIX = 0, int STRAT = -1, bool ONEB = false>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave, (NFIX != 0 && XM <= 1) ? SX_DE_NFIX_WAVES
                                                       : (NFIX == 0 && XM == 0 && STRAT >= 0 && LPR == kWave) ? SX_DE_WAVE_ROWS_WAVES
                                                                                                                 : 1) void de_generation_kernel(const sx_state *const sin_pre,
                                                                                 const double *const pf_pre,
                                                                                 const int64_t *const pi_pre,
                                                                                 const int64_t npart,
                                                                                 const sx_de_args a,
                                                                                 const PlanArg plan,
                                                                                 const int chain_p, const int mode,
                                                                                 const sx_xchg_args x) {
    // sin_pre / pf_pre / pi_pre (single-GPU chained kernel): a.state + chain_p and the record set part[chain_p], worked out
    // on the host and passed, with the number of records, as the FIRST arguments so that they can arrive preloaded in SGPRs
    // (-amdgpu-kernarg-preload-count, sx_de_chain.hip): the first trip to memory after the kernel boundary -- state word
    // and records -- then does not wait for a load of the kernel-argument segment (a by-value struct is never preloaded)
    constexpr bool CHAIN = XM >= 1;  // finalise the predecessor's generation in the prologue
    constexpr bool P2P = XM == 2;    // ... over all ranks, through the peer exchange buffers
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sf[kMaxRowsPerBlock];
    __shared__ int64_t si[kMaxRowsPerBlock];
    __shared__ double s_gf;  // P2P: what wavefront 0 found in the record headers
    __shared__ int64_t s_gi;
    __shared__ int s_winner;
    SX_TP(0);
    const int n = NFIX ? NFIX : a.n;
    const int64_t P = a.P, ld = a.ld;
    const RowIds<LPR> id(P, P2P ? 1 : 0);
    const int l = id.l;  // lane within the row
    const int64_t rowc = id.rowc;
    double *U = lds + id.slot * gen_row_stride(n);

    // ---- A. which generation?  (CHAIN) every wave fetches the predecessor's records right away (their
    //      addresses do not depend on the state word) and keeps them in registers until stage C
    const sx_state *sin = (CHAIN && !P2P) ? sin_pre : CHAIN ? a.state + chain_p : a.state;
    constexpr int kRecPerLane = 8;  // npart <= 512 in chained mode
    double pfv[kRecPerLane];
    // whole-wave rows keep the record rows as 32 bits (rows < 2^31, check_args): 8 registers less is what
    // lets that kernel run 4 waves per SIMD; the short-row kernels are faster with the 64-bit form
    using rec_t = typename std::conditional<LPR == kWave, int32_t, int64_t>::type;
    constexpr rec_t kNoRec = LPR == kWave ? (rec_t)INT32_MAX : (rec_t)INT64_MAX;
    rec_t piv[kRecPerLane];
#ifndef SX_CHAIN_REC_WAVE0
#define SX_CHAIN_REC_WAVE0 1  // wavefront 0 alone reads the records and shares the result through LDS (round 4: all eight
                              // wavefronts reading them was 8 MB of L2 reads per generation at the metric shape: 6.83 -> 6.58 us); 0: A/B
#endif
    if (CHAIN && !P2P && (!SX_CHAIN_REC_WAVE0 || id.wave == 0)) {
        const double *pf = pf_pre;
        const int64_t *pi = pi_pre;
        const int per = (int)((npart + kWave - 1) / kWave);  // contiguous slice per lane: first-minimum rule
        const int64_t k0 = (int64_t)id.lane * per;
#pragma unroll
        for (int u = 0; u < kRecPerLane; ++u) {
            const bool in = u < per && k0 + u < npart;
            pfv[u] = in ? pf[k0 + u] : __builtin_huge_val();
            piv[u] = in ? (rec_t)pi[k0 + u] : kNoRec;
        }
    }
    if (P2P && blockIdx.x == 0) {  // reads the state word itself, after it has issued its record loads
        p2p_service(a, x, chain_p, mode, npart, sin, LPR == kWave);
        return;
    }
    if (sin->done) {
        if (CHAIN && mode == 1 && threadIdx.x == 0) a.state[2] = *sin;
        return;
    }
    // (P2P: an earlier timed-out wait -- *x.error -- is looked at by wavefront 0 in stage C, so that the whole
    //  workgroup takes the same way out and nobody is left alone at a barrier)
    const int64_t it = CHAIN ? sin->it + 1 : sin->it;  // the generation the population holds; we produce it+1
    SX_TP(6);

    // ---- B. everything that only needs `it`: donors, the first batch of row loads, the first Philox call
    const uint32_t gen = (uint32_t)(it + 1);
    const double *__restrict__ cur = (it & 1) ? a.buf1 : a.buf0;
    double *__restrict__ nxt = (it & 1) ? a.buf0 : a.buf1;
    const double fold = a.fit[rowc];
    const double *__restrict__ xi = cur + rowc * ld;
    const uint32_t grow = (uint32_t)(a.row0 + rowc);
    const int strategy = STRAT >= 0 ? STRAT : a.strategy;
    const int k = donors_of(strategy);
    const bool repair = STRAT >= 0 ? false : a.constraints != 0;  // the per-strategy kernels are the constraints=None ones
    const bool use_best = strategy == SX_DE_BEST1BIN || strategy == SX_DE_BEST2BIN;

    int64_t d[kMaxDonors];
    int irand;
    // P2P with global donors: drawn over the WHOLE population (global row ids), fetched from the owner's HBM
    const bool gdon = P2P && x.global_rows > 0;
    if (RNG == SX_RNG_PHILOX) {
        philox_donors(gdon ? x.global_rows : P, k, gdon ? a.row0 + rowc : rowc, grow, gen, a.key0, a.key1, n, d, irand);
    } else {
#pragma unroll
        for (int t = 0; t < kMaxDonors; ++t) d[t] = t < k ? (int64_t)a.donors[(int64_t)t * P + rowc] : 0;
        irand = a.irand[rowc];
    }
    SX_TP(1);
    const double *pd[kMaxDonors];
#pragma unroll
    for (int t = 0; t < kMaxDonors; ++t) pd[t] = cur + d[t] * ld;
    if (gdon) {
#pragma unroll
        for (int t = 0; t < kMaxDonors; ++t) {
            const int owner = (int)(d[t] / x.shard_rows);
            const int64_t lrow = d[t] - (int64_t)owner * x.shard_rows;
            const double *base = nullptr;
#pragma unroll
            for (int r = 0; r < SX_MAX_PEERS; ++r)  // select chain: the kernel-argument arrays stay in SGPRs
                if (r == owner) base = (it & 1) ? x.pop1[r] : x.pop0[r];
            pd[t] = base + lrow * ld;
        }
    }
    const double F = a.F, CR = a.CR;
    const double *r1row = RNG == SX_RNG_HOST ? a.r1 + rowc * (int64_t)n : nullptr;
    const double *rsrow = (RNG == SX_RNG_HOST && repair) ? a.resample + rowc * (int64_t)n : nullptr;

    struct Batch {
        double x[kStep], d[kMaxDonors][kStep], r[kStep], rs[kStep];
    };
    Batch B0;  // (short rows: the row's only batch; whole-wave rows: the batch at work)
    auto load_batch = [&](int q0, Batch &bt) {
        double(&bx)[kStep] = bt.x;
        double(&bd)[kMaxDonors][kStep] = bt.d;
        double(&br)[kStep] = bt.r;
        double(&brs)[kStep] = bt.rs;
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            const bool in = FULL || e < n;
            bx[t] = in ? xi[e] : 0.0;
#pragma unroll
            for (int s = 0; s < kMaxDonors; ++s) {
                if (P2P && gdon)  // a peer's row of the previous generation: read past the caches (system scope)
                    bd[s][t] = (s < k && in) ? __hip_atomic_load(pd[s] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                             : 0.0;
                else
                    bd[s][t] = (s < k && in) ? pd[s][e] : 0.0;
            }
            br[t] = 2.0;
            brs[t] = 0.0;
            if (RNG == SX_RNG_HOST && in) {
                br[t] = r1row[e];
                if (repair) brs[t] = rsrow[e];
            }
        }
        if (RNG == SX_RNG_PHILOX) {
            // 53-bit crossover uniforms, as the reference's rand(P, n) draws them (de/_de.py:250): two per call
            // (slot = (q>>1)*LPR + l, half = q&1), two calls per 4 steps.  (Rounds 1-5 drew 32-bit ones, one call per 4 steps:
            // +0.3 % at M, +2.4 % at C2 for the reference's own precision -- profiles/r6_philox53.txt.)
#pragma unroll
            for (int t = 0; t < kStep; t += 2) {
                const U4 w = philox4x32_10((uint32_t)((q0 + t) >> 1) * (uint32_t)LPR + (uint32_t)l, grow, gen, kPurposeDeCross,
                                           a.key0, a.key1);
                br[t] = u53(w.x, w.y);
                br[t + 1] = u53(w.z, w.w);
            }
            if (repair) {
#pragma unroll
                for (int t = 0; t < kStep; ++t) {
                    const int e = (q0 + t) * LPR + l;
                    if (FULL || e < n)  // np.random.uniform(lo, hi): lo + (hi-lo)*double
                        brs[t] = a.lower[e] + (a.upper[e] - a.lower[e]) *
                                                  philox_u53(e, LPR, grow, gen, kPurposeDeResample, a.key0, a.key1);
                }
            }
        }
    };
    // remote donor rows may only be read once their owners have finished the previous generation, which the
    // arrival of every rank's record (stage C) proves: with global donors the first batch waits for that
    if (!gdon) load_batch(0, B0);

    // ---- C. (CHAIN) best of the predecessor generation, status, publication
    int64_t gbidx = 0;
    double *part_f_out = a.part_f;
    int64_t *part_i_out = a.part_i;
    const uint32_t xtag = (uint32_t)(it + 1);
    const uint64_t *gbw = nullptr;  // P2P: the winner's row as tagged words
    if (P2P) {
        // ONE wavefront per workgroup polls the uncached exchange buffer (all waves doing so would queue 8x the
        // requests on the few channels that hold the headers); the others meet it at the barrier, their donor /
        // row loads already in flight
        const uint64_t *rel = x.relay + (int64_t)chain_p * xchg_relay_stride(n);
        if (id.wave == 0) {
            double bf;
            int64_t bi;
            int winner;
            bool ok = __atomic_load_n(x.error, __ATOMIC_RELAXED) == 0 &&  // an earlier wait timed out: the run is dead
                      xchg_wait_best(x.peer[x.rank] + xchg_slot_offset(n, chain_p, 0), n, x.world, xtag,
                                     x.timeout_ticks, id.lane, bf, bi, winner);
            if (ok && LPR == kWave && use_best && !(it >= 2 && (bf <= a.ftol || it >= a.maxiter))) {
                // whole-wave rows read the winner's row from the cacheable relay: wait (past the caches) until
                // workgroup 0 has published this generation's copy.  Relaxed: the ready word only keeps early
                // readers from caching lines of the previous copy; every word read afterwards is verified by its
                // own tag, so no acquire (= L2 invalidate) is needed
                const uint64_t t0 = wall_clock64();
                while (__hip_atomic_load(rel + xchg_slot_words(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                       (uint64_t)xtag) {
                    if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (id.lane == 0) {
                s_gf = bf;
                s_gi = bi;
                s_winner = ok ? winner : -1;
                if (!ok) atomicExch(x.error, 1);
            }
        }
        __syncthreads();
        const int winner = s_winner;
        if (winner < 0) return;
        const double bf = s_gf;
        const int64_t bi = s_gi;
        if (it >= 2 && (bf <= a.ftol || it >= a.maxiter)) return;  // same rule as the service workgroup
        gbidx = bi;
        // short rows read the winner's row straight from the exchange buffer; whole-wave rows from the relay
        gbw = LPR == kWave ? rel + 4 : x.peer[x.rank] + xchg_slot_offset(n, chain_p, winner) + 4;
        part_f_out = a.part_f + (int64_t)(1 - chain_p) * npart;
        part_i_out = a.part_i + (int64_t)(1 - chain_p) * npart;
        if (gdon) load_batch(0, B0);
    } else if (CHAIN) {
        // (value, first row) over the records: the minimum value first (a tree per lane, then DPP over the wave), then
        // the first record that holds it -- lanes own contiguous slices, so that is the first matching record of the
        // first matching lane.  (A lexicographic compare-and-select chain over the 8 records is ~60 dependent
        // instructions in front of the best row's address.)
        double bf = 0.0;
        int64_t bi = 0;
        if (!SX_CHAIN_REC_WAVE0 || id.wave == 0) {
        double lm[kRecPerLane / 2];
#pragma unroll
        for (int u = 0; u < kRecPerLane / 2; ++u) lm[u] = fmin(pfv[2 * u], pfv[2 * u + 1]);
        const double lmin = fmin(fmin(lm[0], lm[1]), fmin(lm[2], lm[3]));
        bf = wave_min_f64(lmin);
        rec_t br = piv[0];  // (all NaN: record 0, as a sequential scan would)
#pragma unroll
        for (int u = kRecPerLane - 1; u >= 0; --u)
            if (pfv[u] == bf) br = piv[u];
        const unsigned long long hit = __ballot(lmin == bf);
        const int src = hit ? (int)__ffsll((long long)hit) - 1 : 0;
        if (LPR == kWave) {
            bi = (int64_t)__builtin_amdgcn_readlane((int)br, src);
        } else {
            const int lo = __builtin_amdgcn_readlane((int)((int64_t)br & 0xffffffffll), src);
            const int hi = __builtin_amdgcn_readlane((int)((int64_t)br >> 32), src);
            bi = ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
        }
        }
        if (SX_CHAIN_REC_WAVE0) {
            if (id.wave == 0 && id.lane == 0) s_gf = bf, s_gi = bi;
            __syncthreads();
            bf = s_gf, bi = s_gi;
        }
        int status = SX_STATUS_NONE;
        if (it >= 2) {  // the reference does not test the initial population (de/_de.py:212-218)
            if (bf <= a.ftol)
                status = 1;
            else if (it >= a.maxiter)
                status = -1;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            sx_state *so = a.state + (mode == 1 ? 2 : 1 - chain_p);
            so->it = it;
            so->gbidx = bi;
            so->gfit = bf;
            so->dx = 0.0;
            so->status = status;
            so->done = status != SX_STATUS_NONE;
            so->reserved[0] = sin->gbidx;
        }
        if (status != SX_STATUS_NONE || mode == 1) return;
        gbidx = bi;
        part_f_out = a.part_f + (int64_t)(1 - chain_p) * npart;
        part_i_out = a.part_i + (int64_t)(1 - chain_p) * npart;
    } else {
        gbidx = sin->gbidx;
    }
    // best row: the caller's copy (multi-GPU: it may come from another shard) or row gbidx of this generation
    const double *__restrict__ gb = P2P ? cur : a.gbest != nullptr ? a.gbest : cur + gbidx * ld;

    // ---- D. trial vector: mutation (de/_strategy.py, same association), crossover (de/_de.py:344 forced
    //      index OR r <= CR), Random repair (de/_constraints.py:21-26) -> LDS
    const int nq = (n + LPR - 1) / LPR;
    // one batch per row -- NFIX, and (round 5) EVERY row of a short-row kernel, whatever its run-time length (lanes_per_row
    // gives rows of up to 64 / 128 elements 16 / 32 lanes: n <= kStep * LPR): trial and own row stay in registers for the
    // row store
    constexpr bool kRegRow = NFIX != 0 || LPR < kWave || ONEB;
    double keep[kStep];
    auto trial_batch = [&](int q0, const Batch &bt) {
        const double(&bx)[kStep] = bt.x;
        const double(&bd)[kMaxDonors][kStep] = bt.d;
        const double(&br)[kStep] = bt.r;
        const double(&brs)[kStep] = bt.rs;
        double g[kStep];
        if (P2P) {
            if (use_best) {  // tagged words from the exchange buffer; a word not yet there is simply re-read
                const uint64_t t0 = wall_clock64();
                for (int attempt = 0;; ++attempt) {
                    uint64_t lo[kStep], hi[kStep];
                    bool ok = true;
                    // relay (LPR = 64): first an ordinary, cacheable load -- a stale or not yet written word
                    // fails the tag test and is then re-read past the caches (device scope) until it is there
                    const bool cached = LPR == kWave && attempt == 0;
#pragma unroll
                    for (int t = 0; t < kStep; ++t) {
                        const int e = (q0 + t) * LPR + l;
                        const bool in = FULL || e < n;
                        if (cached) {  // both tagged halves of a double in one 16-byte load
                            const ulonglong2 w2 =
                                in ? *reinterpret_cast<const ulonglong2 *>(gbw + 2 * e) : make_ulonglong2(0, 0);
                            lo[t] = w2.x;
                            hi[t] = w2.y;
                        } else if (LPR == kWave) {
                            lo[t] = in ? __hip_atomic_load(gbw + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                            hi[t] = in ? __hip_atomic_load(gbw + 2 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                        } else {
                            lo[t] = in ? ll_load(gbw + 2 * e) : 0;
                            hi[t] = in ? ll_load(gbw + 2 * e + 1) : 0;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < kStep; ++t) {
                        const int e = (q0 + t) * LPR + l;
                        if (FULL || e < n) ok = ok && ll_ok(lo[t], xtag) && ll_ok(hi[t], xtag);
                        g[t] = ll_join_f64(lo[t], hi[t]);
                    }
                    if (ok) break;
                    if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
                        atomicExch(x.error, 1);
                        break;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < kStep; ++t) g[t] = 0.0;
            }
        } else {
#pragma unroll
            for (int t = 0; t < kStep; ++t) {
                const int e = (q0 + t) * LPR + l;
                g[t] = (use_best && (FULL || e < n)) ? gb[e] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < kStep; ++t) {
            const int e = (q0 + t) * LPR + l;
            if (FULL || e < n) {
                const double v = de_mutant(strategy, g[t], bd[0][t], bd[1][t], bd[2][t], bd[3][t], bd[4][t], F);
                double cand = (e == irand || br[t] <= CR) ? v : bx[t];
                if (repair && (cand < a.lower[e] || cand > a.upper[e])) cand = brs[t];
                U[e] = cand;
                if (kRegRow) keep[t] = cand;
            }
        }
    };
    if constexpr (LPR < kWave) {
        // short rows are ONE batch (lanes_per_row gives 16 / 32 lanes to rows of up to 64 / 128 elements only).  Round 5: said
        // at compile time -- the general short-row kernels carried a second batch's 64 registers for a loop that never took a
        // second trip (185 VGPRs: two wavefronts per SIMD)
        trial_batch(0, B0);
    } else {  // whole-wave rows (n > 128): one batch and half the registers -- twice the waves hide it better
        trial_batch(0, B0);
        if constexpr (!ONEB) {
            for (int q0 = kStep; q0 < nq; q0 += kStep) {
                load_batch(q0, B0);
                trial_batch(q0, B0);
            }
        }
    }

    SX_TP(2);
    const double fc = row_objective<FUN, LPR, FULL, NFIX, SX_LONG_STATIC, ONEB>(U, n, plan, l);
    SX_TP(3);
    const bool better = fc < fold;  // _common.py:127 strict <
    if (FULL || id.active) {
        double *__restrict__ xo = nxt + id.row * ld;
        if constexpr (kRegRow) {  // both rows are in registers already: no load behind the objective; streaming stores
#pragma unroll
            for (int t = 0; t < kStep; ++t)
                if (NFIX != 0 || l + t * LPR < n) st_stream(xo + l + t * LPR, better ? keep[t] : B0.x[t]);
        } else {
            const double *__restrict__ src = better ? U : xi;  // LDS or global: generic loads
            for (int e0 = l; e0 < n; e0 += kStep * LPR) {
                double v[kStep];
#pragma unroll
                for (int t = 0; t < kStep; ++t) v[t] = (FULL || e0 + t * LPR < n) ? src[e0 + t * LPR] : 0.0;
#pragma unroll
                for (int t = 0; t < kStep; ++t)
                    if (FULL || e0 + t * LPR < n) xo[e0 + t * LPR] = v[t];
            }
        }
        if (l == 0) {
            if (better) a.fit[id.row] = fc;
            if (a.candfit != nullptr) a.candfit[id.row] = fc;
        }
    }
    SX_TP(4);
    block_partial<LPR>(better ? fc : fold, id, sf, si, part_f_out, part_i_out);
    SX_TP(5);
}

typedef void (*de_kernel_t)(const sx_state *, const double *, const int64_t *, const int64_t, const sx_de_args,
                            const PlanArg, const int, const int, const sx_xchg_args);

template <int RNG, int XM, int LPR, bool FULL, int NFIX = 0, int STRAT = -1, bool ONEB = false>
de_kernel_t pick_kernel_lpr(int fun_id) {
    constexpr bool SPECIAL = STRAT != -1 || ONEB || NFIX != 0;  // (callers route the other objectives to the general form: hot_objective)
    switch (fun_id) {
        case SX_FUN_ACKLEY: return de_generation_kernel<SX_FUN_ACKLEY, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
        case SX_FUN_RASTRIGIN: return de_generation_kernel<SX_FUN_RASTRIGIN, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
        case SX_FUN_ROSENBROCK: return de_generation_kernel<SX_FUN_ROSENBROCK, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
        case SX_FUN_SPHERE: return de_generation_kernel<SX_FUN_SPHERE, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
    }
    if constexpr (!SPECIAL) {
        switch (fun_id) {
            case SX_FUN_GRIEWANK: return de_generation_kernel<SX_FUN_GRIEWANK, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
            case SX_FUN_QUARTIC: return de_generation_kernel<SX_FUN_QUARTIC, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
            case SX_FUN_STYBLINSKI_TANG: return de_generation_kernel<SX_FUN_STYBLINSKI_TANG, RNG, XM, LPR, FULL, NFIX, STRAT, ONEB>;
        }
    }
    return nullptr;
}

// bounds-test-free variant only for the chained (throughput) kernels, to keep the build small
// the one-batch kernels come per strategy
template <int RNG, int XM, int LPR, bool FULL, int NFIX>
de_kernel_t pick_kernel_fixed(int fun_id, int strategy, int constraints) {
    // per strategy for the single-GPU chained kernel; for the peer-exchange kernel (whose instantiations are the
    // expensive ones to compile) only at the metric shape's row length, n = 128 -- what `bench.py --gpus N` runs
    if constexpr (XM == 2 && LPR != 32) {
        return pick_kernel_lpr<RNG, XM, LPR, FULL, NFIX>(fun_id);
    } else {
        if (constraints != 0 || !hot_objective(fun_id)) return pick_kernel_lpr<RNG, XM, LPR, FULL, NFIX>(fun_id);
        switch (strategy) {  // (rand2bin, best2bin: the run-time strategy of the same kernel)
            case SX_DE_RAND1BIN: return pick_kernel_lpr<RNG, XM, LPR, FULL, NFIX, NFIX ? SX_DE_RAND1BIN : -1>(fun_id);
            case SX_DE_BEST1BIN: return pick_kernel_lpr<RNG, XM, LPR, FULL, NFIX, NFIX ? SX_DE_BEST1BIN : -1>(fun_id);
        }
        return pick_kernel_lpr<RNG, XM, LPR, FULL, NFIX>(fun_id);
    }
}

// the general (run-time row length) kernel with the strategy as a compile-time constant: single-GPU forms with in-kernel draws
// (XM = 0: the two-kernel path, what populations of more than 8192 rows take; XM = 1: the chained kernel), and only for the two
// strategies everyone uses.  Round 5:
// with five donor slots and the repair code compiled in, the general short-row kernel takes 89 VGPRs -- two 512-thread
// workgroups per CU; best1bin alone needs 2 donors: <= 80 VGPRs, three workgroups (what bounds these shapes is latency:
// profiles/r5_de_gather_probe.txt)
template <int RNG, int XM, int LPR>
de_kernel_t pick_kernel_general(int fun_id, int strategy, int constraints, int n) {
    if constexpr (XM <= 1 && RNG == SX_RNG_PHILOX) {
        if (!hot_objective(fun_id)) return pick_kernel_lpr<RNG, XM, LPR, false>(fun_id);
        if constexpr (LPR == kWave) {  // whole-wave rows of up to 256 elements: the one-batch form (ONEB)
            if (n <= 4 * kWave && constraints == 0 && strategy == SX_DE_BEST1BIN)
                return pick_kernel_lpr<RNG, XM, LPR, false, 0, SX_DE_BEST1BIN, true>(fun_id);
            if (n <= 4 * kWave && constraints == 0 && strategy == SX_DE_RAND1BIN)
                return pick_kernel_lpr<RNG, XM, LPR, false, 0, SX_DE_RAND1BIN, true>(fun_id);
        }
        if (constraints == 0 && strategy == SX_DE_BEST1BIN) return pick_kernel_lpr<RNG, XM, LPR, false, 0, SX_DE_BEST1BIN>(fun_id);
        if (constraints == 0 && strategy == SX_DE_RAND1BIN) return pick_kernel_lpr<RNG, XM, LPR, false, 0, SX_DE_RAND1BIN>(fun_id);
    }
    return pick_kernel_lpr<RNG, XM, LPR, false>(fun_id);
}

template <int RNG, int XM>
de_kernel_t pick_kernel(int fun_id, int n, int64_t P, int strategy, int constraints) {
    constexpr bool CH = XM >= 1;
    const int lpr = lanes_per_row(n);
    const bool full = CH && n % (kStep * lpr) == 0 && P % rows_per_block(n) == 0;
    // one batch per row exactly (n = 64, 128, 256) with in-kernel draws: the compile-time row length
    constexpr bool FX = CH && RNG == SX_RNG_PHILOX;
    const bool fix = FX && full && n == kStep * lpr && hot_objective(fun_id);  // (the others: the run-time row length)
    switch (lpr) {
        case 16:
            if (fix) return pick_kernel_fixed<RNG, XM, 16, CH, FX ? kStep * 16 : 0>(fun_id, strategy, constraints);
            return full ? pick_kernel_lpr<RNG, XM, 16, CH>(fun_id) : pick_kernel_general<RNG, XM, 16>(fun_id, strategy, constraints, n);
        case 32:
            if (fix) return pick_kernel_fixed<RNG, XM, 32, CH, FX ? kStep * 32 : 0>(fun_id, strategy, constraints);
            return full ? pick_kernel_lpr<RNG, XM, 32, CH>(fun_id) : pick_kernel_general<RNG, XM, 32>(fun_id, strategy, constraints, n);
    }
    if (fix) return pick_kernel_fixed<RNG, XM, 64, CH, FX ? kStep * 64 : 0>(fun_id, strategy, constraints);
    return full ? pick_kernel_lpr<RNG, XM, 64, CH>(fun_id) : pick_kernel_general<RNG, XM, 64>(fun_id, strategy, constraints, n);
}

}  // namespace
