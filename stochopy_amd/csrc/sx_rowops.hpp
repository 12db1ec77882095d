// Row-level device routines shared by the evaluation / DE / PSO kernels:
// row <-> lane indexing, LDS-staged objective evaluation, per-workgroup best.
#pragma once
#include "sx_device.hpp"

namespace sx {

// LPR lanes own one row; a wave carries RPW = 64 / LPR rows.
template <int LPR>
struct RowIds {
    static constexpr int RPW = kWave / LPR;
    int wave, lane;  // wave in workgroup, lane in wave
    int l;           // lane within the row
    int slot;        // row slot within the workgroup
    int block;       // row workgroup index (blockIdx.x minus the leading service workgroups)
    int64_t row, rowc;
    bool active;
    __device__ __forceinline__ explicit RowIds(int64_t P, int first_block = 0) {
        wave = (int)(threadIdx.x >> 6);
        lane = (int)(threadIdx.x & 63);
        l = lane & (LPR - 1);
        slot = wave * RPW + lane / LPR;
        const int rows_in_block = (int)(blockDim.x >> 6) * RPW;
        block = (int)blockIdx.x - first_block;
        row = (int64_t)block * rows_in_block + slot;
        active = row < P;
        rowc = active ? row : P - 1;  // padding rows shadow the last row and store nothing
    }
};

struct Geometry {
    unsigned blocks, threads;
    size_t lds;
};
inline Geometry row_geometry(int64_t P, int n) {
    // wide rows (n > kWideFrom, sx_wide.hip): one row and one record per workgroup; the kernels that still walk such a row with
    // ONE wavefront (candidate / selection / radius kernels: no staging) are launched with 64 threads and no dynamic LDS
    if (n > sx::wide_from()) return Geometry{(unsigned)P, (unsigned)kWave, 0};
    const int wpb = waves_per_block(n);
    const int rpb = rows_per_block(n);
    return Geometry{(unsigned)((P + rpb - 1) / rpb), (unsigned)(wpb * kWave),
                    (size_t)rpb * lds_row_stride(n) * sizeof(double)};
}

// LDS traffic of one wavefront is processed in issue order, so data a lane wrote is
// visible to the other lanes of the SAME wave once the write has been issued; this
// only keeps the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void lds_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reductions over the LPR lanes of a row (LPR is a power of two, rows are LPR-aligned)
template <int LPR>
__device__ __forceinline__ double row_min(double v) {
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) v = fmin(v, __shfl_xor(v, off, kWave));
    return v;
}
template <int LPR>
__device__ __forceinline__ double row_sum(double v) {
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// One-batch rows of 64 / 128 elements (one leaf of numpy's recursion: <= 128 terms), round 4: the objective WITHOUT the
// term arrays.  numpy's sum is eight accumulators r_j = a[j] + a[8+j] + a[16+j] + ... (strictly left to right), the tree
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail terms one by one (loops_utils.h.src pairwise_sum).  Lane
// L = SEG*j + g of the row (SEG = LPR/8 lanes per accumulator) reads the four elements 8(4g+i)+j, i = 0..3, of the staged
// vector, forms their terms in registers and adds them -- in order -- onto the running sum that arrives from lane L-1
// (same quad: one DPP move): accumulator j's chain walks through SEG adjacent lanes and ends in lane SEG*j + SEG-1.
// Same additions in the same order as row_reduce_static / row_reduce_fixed (same bits), but per row 2 LDS reads per term
// instead of two staged term arrays written and read back by every 8-lane group (the objective was 0.7 us of the 6.9 us
// generation at the metric shape).
template <int FUN, int LPR, int NFIX>
__device__ __forceinline__ double row_objective_chain(const double *U, int l) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    // n = 256 with one term per element: numpy splits 256 = 128 + 128, two such leaves -- the row's two 32-lane halves
    constexpr int NLEAF = NFIX > 128 ? 2 : 1;
    constexpr int M = (O::NEXT ? NFIX - 1 : NFIX) / NLEAF;  // terms per leaf
    constexpr int SEG = LPR / kGroup / NLEAF;               // lanes per accumulator chain: 2 (n = 64) or 4 (n = 128, 256)
    static_assert(NFIX == 4 * LPR && (SEG == 2 || SEG == 4) && M <= 128 && (NLEAF == 1 || !O::NEXT),
                  "one leaf (or two full ones), four elements per lane");
    constexpr int NB = M / kGroup, TAIL = M % kGroup;  // blocks that go into the accumulators; tail terms
    static_assert(TAIL == 0 || NB == 4 * SEG - 1, "the tail is the last block");
    constexpr int kHop = SEG == 4 ? 0x90 : 0xA0;  // quad_perm [0,0,1,2] / [0,0,2,2]: lane L takes lane L-1's value
    constexpr int LPL = LPR / NLEAF;               // lanes per leaf
    const int leaf0 = NLEAF == 2 ? (l / LPL) * (NFIX / 2) : 0;  // first element of this lane's leaf
    const int j = (l % LPL) / SEG, g = l % SEG;
    double a[4], b[4];
    {
        double x[4], xn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = leaf0 + kGroup * (4 * g + i) + j;
            x[i] = U[e];
            xn[i] = O::NEXT ? U[e + 1] : 0.0;  // (U[NFIX .. NFIX+7] is padding: the term of element NFIX-1 is never used)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) O::term(x[i], xn[i], leaf0 + kGroup * (4 * g + i) + j, a[i], b[i]);
    }
    // the chain: segment s continues from what segment s-1 hands over.  Every lane runs every segment's code; only the
    // lanes whose turn it is (g == s) hold meaningful values, and only those are passed on.
    double ra = a[0], rb = b[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (i < NB) {
            ra = ra + a[i];
            if (TWO) rb = combine<BMUL>(rb, b[i]);
        }
    }
#pragma unroll
    for (int s = 1; s < SEG; ++s) {
        ra = dpp_f64<kHop>(ra);
        if (TWO) rb = dpp_f64<kHop>(rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (4 * s + i < NB) {
                ra = ra + a[i];
                if (TWO) rb = combine<BMUL>(rb, b[i]);
            }
        }
    }
    // r_j sits in lane SEG*j + SEG-1; the tree over j (each + / * commutes exactly)
    constexpr int kShl1 = 0x100 + SEG, kShl2 = 0x100 + 2 * SEG;  // row_shl: lane L takes lane L+SEG / L+2 SEG
    ra = combine<false>(ra, dpp_f64<kShl1>(ra));
    ra = combine<false>(ra, dpp_f64<kShl2>(ra));
    if (TWO) {
        rb = combine<BMUL>(rb, dpp_f64<kShl1>(rb));
        rb = combine<BMUL>(rb, dpp_f64<kShl2>(rb));
    }
    double sa, sb = BMUL ? 1.0 : 0.0;
    if constexpr (SEG == 2) {  // a row is one 16-lane DPP row: the last level is a DPP move too
        ra = combine<false>(ra, dpp_f64<0x108>(ra));
        if (TWO) rb = combine<BMUL>(rb, dpp_f64<0x108>(rb));
        sa = row_lane_value<LPR>(ra, SEG - 1, l);
        if (TWO) sb = row_lane_value<LPR>(rb, SEG - 1, l);
    } else if constexpr (NLEAF == 1) {  // two DPP rows: (r0..r3) in lane 3, (r4..r7) in lane 19
        sa = row_lane_value<LPR>(ra, SEG - 1, l) + row_lane_value<LPR>(ra, 4 * SEG + SEG - 1, l);
        if (TWO) sb = combine<BMUL>(row_lane_value<LPR>(rb, SEG - 1, l), row_lane_value<LPR>(rb, 4 * SEG + SEG - 1, l));
    } else {  // two leaves: leaf 0 ends in lanes 3 / 19, leaf 1 in lanes 35 / 51; numpy adds leaf 0 + leaf 1
        sa = (row_lane_value<LPR>(ra, 3, l) + row_lane_value<LPR>(ra, 19, l)) +
             (row_lane_value<LPR>(ra, 35, l) + row_lane_value<LPR>(ra, 51, l));
        if (TWO)
            sb = combine<BMUL>(combine<BMUL>(row_lane_value<LPR>(rb, 3, l), row_lane_value<LPR>(rb, 19, l)),
                               combine<BMUL>(row_lane_value<LPR>(rb, 35, l), row_lane_value<LPR>(rb, 51, l)));
    }
    if constexpr (TAIL > 0) {
        // the tail terms a[M-7 .. M-1] are added one by one to the total: every lane forms them itself from TAIL + 1
        // broadcast reads of the staged vector (independent of the chains above, so they hide behind them) instead of
        // fetching them from the lanes that own those elements (4 cross-lane moves per term and stream)
        double tx[TAIL + 1];
#pragma unroll
        for (int t = 0; t <= TAIL; ++t) tx[t] = U[kGroup * NB + t];
#pragma unroll
        for (int t = 0; t < TAIL; ++t) {
            double ta, tb;
            O::term(tx[t], tx[t + 1], kGroup * NB + t, ta, tb);
            sa = sa + ta;
            if (TWO) sb = combine<BMUL>(sb, tb);
        }
    }
    sa = 0.0 + sa;  // add.reduce starts from the identity
    sb = !TWO ? (BMUL ? 1.0 : 0.0) : (BMUL ? sb : 0.0 + sb);
    return O::finish(sa, sb, NFIX);
}

// objectives whose terms are a few multiplications (no cosine): for these the summation plan, not the arithmetic, is the
// row's time, and long rows of the usual lengths get the plan as constants (row_reduce_long)
template <int FUN>
constexpr bool light_objective() {
    return FUN == SX_FUN_ROSENBROCK || FUN == SX_FUN_SPHERE || FUN == SX_FUN_QUARTIC || FUN == SX_FUN_STYBLINSKI_TANG;
}
// The same register chains for rows whose length is only known at run time (round 5: shapes off the benchmark grid -- n = 100,
// 130, 250 ... -- used to take the staged-terms path: term arrays written to LDS, the plan walked with scalar loads).  A row
// that fits one batch (n <= 4 LPR, i.e. EVERY row of up to 256 elements) has at most three leaves in numpy's recursion, each
// of at most 16 blocks, and only the last one has a tail:
//   m <= 128: one leaf;   m in 129..256: split at n2 = (m/2) - (m/2) % 8 -- and the right part r = m - n2 once more if it
//   is still above 128 (129..135: m = 249..255): 64 + (r - 64), two leaves of at most 8 blocks -- S0 + (S1 + S2).
// A leaf of nb blocks from element e0 lives on a group of 8 SEG lanes exactly as in row_objective_chain -- lane SEG j + g forms
// the terms of blocks 4g .. 4g+3 of accumulator j (leaf_terms_rt: every lane at most four terms, whatever the leaf layout), the
// running sum hops from lane to lane by DPP (leaf_chain_rt) -- with the block count as a run-time value: a block beyond it
// contributes nothing.  After leaf_chain_rt lane SEG-1 (SEG = 2: the whole tree) or lanes SEG-1 and 4 SEG + SEG-1 (SEG = 4:
// the two halves of the last tree level) of the group hold the leaf's sums.  Same additions in the same order as numpy.
template <int FUN>
__device__ __forceinline__ void leaf_terms_rt(const double *U, int e0, int nb, int j, int g, double (&a)[4], double (&b)[4]) {
    using O = Obj<FUN>;
    double x[4], xn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool in = 4 * g + i < nb;
        const int e = e0 + kGroup * (4 * g + i) + j;
        x[i] = in ? U[e] : 0.0;
        xn[i] = (O::NEXT && in) ? U[e + 1] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) O::term(x[i], xn[i], e0 + kGroup * (4 * g + i) + j, a[i], b[i]);
}
template <int FUN, int SEG>
__device__ __forceinline__ void leaf_chain_rt(const double (&a)[4], const double (&b)[4], int nb, double &ra, double &rb) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    static_assert(SEG == 2 || SEG == 4, "16 or 32 lanes per leaf");
    constexpr int kHop = SEG == 4 ? 0x90 : 0xA0;  // quad_perm [0,0,1,2] / [0,0,2,2]: lane L takes lane L-1's value
    ra = a[0], rb = b[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const bool in = i < nb;
        ra = in ? ra + a[i] : ra;
        if (TWO) rb = in ? combine<BMUL>(rb, b[i]) : rb;
    }
#pragma unroll
    for (int s = 1; s < SEG; ++s) {
        ra = dpp_f64<kHop>(ra);
        if (TWO) rb = dpp_f64<kHop>(rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = 4 * s + i < nb;
            ra = in ? ra + a[i] : ra;
            if (TWO) rb = in ? combine<BMUL>(rb, b[i]) : rb;
        }
    }
    constexpr int kShl1 = 0x100 + SEG, kShl2 = 0x100 + 2 * SEG;  // row_shl: lane L takes lane L+SEG / L+2 SEG
    ra = combine<false>(ra, dpp_f64<kShl1>(ra));
    ra = combine<false>(ra, dpp_f64<kShl2>(ra));
    if (TWO) {
        rb = combine<BMUL>(rb, dpp_f64<kShl1>(rb));
        rb = combine<BMUL>(rb, dpp_f64<kShl2>(rb));
    }
    if constexpr (SEG == 2) {  // the group is one 16-lane DPP row: the last level is a DPP move too
        ra = combine<false>(ra, dpp_f64<0x108>(ra));
        if (TWO) rb = combine<BMUL>(rb, dpp_f64<0x108>(rb));
    }
}

#ifndef SX_OBJ_CHAIN_RT
#define SX_OBJ_CHAIN_RT 1  // A/B switch: 0 = one-batch rows of run-time length stage their terms (rounds 1-4)
#endif

template <int FUN, int LPR>
__device__ __forceinline__ double row_objective_chain_rt(const double *U, int n, const PlanArg &plan, int l) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    const double identB = BMUL ? 1.0 : 0.0;
    const int m = O::NEXT ? n - 1 : n;
    const int nbt = m / kGroup, tail = m % kGroup;  // blocks in all leaves together; terms behind the last leaf's tree
    // the tail terms a[8 nbt .. m-1] are added one by one to the last leaf's sum.  Cheap terms: every lane forms them itself from
    // broadcast reads; a cosine per term: lane t of the row forms term t, and the sums below fetch it
    constexpr bool kTailByLane = !light_objective<FUN>() || LPR == kWave;
    double tla = 0.0, tlb = identB;
    if (kTailByLane && tail > 0) {
        const int e = kGroup * nbt + (l < tail ? l : 0);
        O::term(U[e], O::NEXT ? U[e + 1] : 0.0, e, tla, tlb);
    }
    auto add_tail = [&](double &xa, double &xb) {
#pragma unroll
        for (int t = 0; t < kGroup - 1; ++t) {
            if (t < tail) {  // (uniform)
                double ta, tb;
                if constexpr (kTailByLane) {
                    ta = row_lane_value<LPR>(tla, t, l);
                    tb = TWO ? row_lane_value<LPR>(tlb, t, l) : identB;
                } else {
                    const int e = kGroup * nbt + t;
                    O::term(U[e], O::NEXT ? U[e + 1] : 0.0, e, ta, tb);
                }
                xa = xa + ta;
                if (TWO) xb = combine<BMUL>(xb, tb);
            }
        }
    };
    double sa = 0.0, sb = identB;
    double a[4], b[4], ra, rb;
    if constexpr (LPR < kWave) {  // n <= 128: at most one leaf, on all of the row's lanes
        constexpr int SEG = LPR / kGroup;
        leaf_terms_rt<FUN>(U, 0, nbt, l / SEG, l % SEG, a, b);
        leaf_chain_rt<FUN, SEG>(a, b, nbt, ra, rb);
        if (nbt > 0) {
            if constexpr (SEG == 2) {
                sa = row_lane_value<LPR>(ra, SEG - 1, l);
                if (TWO) sb = row_lane_value<LPR>(rb, SEG - 1, l);
            } else {
                sa = row_lane_value<LPR>(ra, SEG - 1, l) + row_lane_value<LPR>(ra, 4 * SEG + SEG - 1, l);
                if (TWO) sb = combine<BMUL>(row_lane_value<LPR>(rb, SEG - 1, l), row_lane_value<LPR>(rb, 4 * SEG + SEG - 1, l));
            }
        }
        add_tail(sa, sb);
    } else {
        // whole-wave rows of up to 256 elements.  One or two leaves: leaf 0 on lanes 0..31, leaf 1 on lanes 32..63 (32 lanes
        // each: SEG = 4).  Three leaves: leaf 0 as before, leaves 1 and 2 (at most 8 blocks each) on lanes 32..47 and 48..63
        // (16 lanes each: SEG = 2).  Terms are formed ONCE, four per lane; with three leaves both chain forms run over them and
        // each half of the wave keeps the one that is its own.
        const int nleaf = plan.nleaf;  // 0 (m < 8), 1, 2 or 3
        const int end0 = nleaf > 0 ? plan.end[0] : 0, end1 = nleaf > 1 ? plan.end[1] : end0;
        const bool three = nleaf > 2;  // (uniform)
        const int half = l >> 5;
        const bool narrow = three && half;  // this lane belongs to a 16-lane group
        const int lg = narrow ? (l & 15) : (l & 31), grp = (l >> 4) & 1;
        const int seg = narrow ? 2 : 4;
        const int e0 = kGroup * (!half ? 0 : !three ? end0 : grp ? end1 : end0);
        const int nb = !half ? end0 : !three ? end1 - end0 : grp ? nbt - end1 : end1 - end0;
        leaf_terms_rt<FUN>(U, e0, nb, lg / seg, lg % seg, a, b);
        leaf_chain_rt<FUN, 4>(a, b, nb, ra, rb);
        const double s0a = readlane_f64(ra, 3) + readlane_f64(ra, 19);
        double s1a = readlane_f64(ra, 35) + readlane_f64(ra, 51), s2a = 0.0;
        double s0b = identB, s1b = identB, s2b = identB;
        if (TWO) {
            s0b = combine<BMUL>(readlane_f64(rb, 3), readlane_f64(rb, 19));
            s1b = combine<BMUL>(readlane_f64(rb, 35), readlane_f64(rb, 51));
        }
        if (three) {  // (uniform)
            leaf_chain_rt<FUN, 2>(a, b, nb, ra, rb);
            s1a = readlane_f64(ra, 33), s2a = readlane_f64(ra, 49);
            if (TWO) s1b = readlane_f64(rb, 33), s2b = readlane_f64(rb, 49);
        }
        // the last leaf takes the tail before the merges
        double la = three ? s2a : nleaf > 1 ? s1a : s0a, lb = three ? s2b : nleaf > 1 ? s1b : s0b;
        if (nleaf == 0) la = 0.0, lb = identB;
        add_tail(la, lb);
        // merges in recursion order: S0 + S1, or S0 + (S1 + S2)
        if (three) {
            sa = s0a + (s1a + la);
            if (TWO) sb = combine<BMUL>(s0b, combine<BMUL>(s1b, lb));
        } else if (nleaf > 1) {
            sa = s0a + la;
            if (TWO) sb = combine<BMUL>(s0b, lb);
        } else {
            sa = la, sb = lb;
        }
    }
    sa = 0.0 + sa;  // add.reduce starts from the identity
    sb = !TWO ? identB : (BMUL ? sb : 0.0 + sb);
    return O::finish(sa, sb, n);
}

// rows whose objective needs nothing but the staged vector itself (row_objective_chain): kernels that stage for nobody else
// (sx_eval) can then give a row n + 8 doubles of LDS instead of lds_row_stride(n) and fit twice the workgroups on a CU
#ifndef SX_OBJ_CHAIN
#define SX_OBJ_CHAIN 1  // A/B switch: 0 = the staged-terms form for one-batch rows too
#endif
#ifndef SX_OBJ_CHAIN256
#define SX_OBJ_CHAIN256 1  // A/B: 0 = rows of 256 elements stage their terms
#endif
#ifndef SX_LONG_STATIC
#define SX_LONG_STATIC 1  // (0: long rows inside the generation kernels keep the run-time plan -- A/B builds)
#endif
template <int FUN, int NFIX>
constexpr bool chain_only() {
    return SX_OBJ_CHAIN && NFIX != 0 && (NFIX <= 128 || (SX_OBJ_CHAIN256 && NFIX == 256 && !Obj<FUN>::NEXT));
}

// Objective of the row staged in LDS at U[0..n): terms by the row's LPR lanes -> A/B (behind U),
// then the numpy-order row sums (lanes l >= 8 repeat the chains of lanes l & 7: LDS broadcasts, same bits).
// Every lane of the row returns the value.  Each row works on its own LDS slice (no workgroup barrier).
// NFIX: the row length when it is a compile-time constant (the PSO kernel's whole-batch rows), else 0
// LONGSTATIC = 0: no branch to the compile-time plans of n = 512 / 1024 / 2048 (the one-workgroup kernels of
// updating="immediate", 1 024 threads at the 128-VGPR cap, would spill for them)
// ONEBATCH: the caller guarantees n <= 4 LPR (a whole-wave kernel instantiated for rows of up to 256 elements): nothing but the
// run-time register chain is compiled (the kernel then does not carry the long rows' plans in its register budget)
template <int FUN, int LPR, bool FULL = false, int NFIX = 0, int LONGSTATIC = SX_LONG_STATIC, bool ONEBATCH = false>
__device__ __forceinline__ double row_objective(double *U, int n, const PlanArg &plan, int l) {
    using O = Obj<FUN>;
    const int m = O::NEXT ? n - 1 : n;
    lds_wave_fence();  // U complete (written and read by this wave only)
    if constexpr (chain_only<FUN, NFIX>())
        return row_objective_chain<FUN, LPR, NFIX>(U, l);
    if constexpr ((NFIX == 0 || NFIX <= 256) && SX_OBJ_CHAIN_RT) {  // one-batch rows (every n <= 256) without a compile-time chain form: the same chains, run-time block counts
        // (NFIX = 256 with an objective that reads the next element -- m = 255: three leaves -- comes here too: no kernel stages
        //  terms for a row of up to 256 elements any more, so such rows need n + 8 doubles of LDS, sx_device.hpp de_row_stride)
        // (lanes_per_row gives 16 / 32 lanes to rows of up to 64 / 128 elements only: a short-row kernel never meets a longer row,
        //  and the staged-terms code below is not even compiled for it)
        if constexpr (LPR < kWave || ONEBATCH) return row_objective_chain_rt<FUN, LPR>(U, n, plan, l);
        if (n <= 4 * LPR) return row_objective_chain_rt<FUN, LPR>(U, n, plan, l);
    }
    if constexpr (NFIX > 256) {  // a long row of compile-time length: numpy's plan as constants (row_reduce_long)
        static_assert(LPR == kWave, "whole-wave rows");
        double sa, sb;
        row_reduce_long<FUN, (O::NEXT ? NFIX - 1 : NFIX), (light_objective<FUN>() ? 8 : 4)>(U, l, sa, sb);
        return O::finish(sa, sb, NFIX);
    }
    if constexpr (LONGSTATIC != 0 && NFIX == 0 && LPR == kWave && light_objective<FUN>()) {
        // long rows of the usual lengths inside the generation kernels (BASELINE config 5: n = 1024): the same constants, picked by
        // a uniform branch on the run-time length
        if (n == 1024 || n == 512 || n == 2048) {
            double sa, sb;
            if (n == 1024)
                row_reduce_long<FUN, (O::NEXT ? 1023 : 1024)>(U, l, sa, sb);
            else if (n == 512)
                row_reduce_long<FUN, (O::NEXT ? 511 : 512)>(U, l, sa, sb);
            else
                row_reduce_long<FUN, (O::NEXT ? 2047 : 2048)>(U, l, sa, sb);
            return O::finish(sa, sb, n);
        }
    }
    if (LPR == kWave && fused_terms(n)) {  // terms are formed inside the reduction, nothing else is staged
        double sa, sb;
        row_reduce_leaves_fused<FUN, LPR>(U, U + n + 8, leaf_cap(n), m, plan, l, sa, sb);
        return O::finish(sa, sb, n);
    }
    double *A = U + n + 8;
    double *B = A + n;
    for (int e0 = l; e0 < m; e0 += 4 * LPR) {  // 4 steps per trip: the LDS reads of a trip are independent
        double x[4], xn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = e0 + t * LPR;
            // FULL: rows are whole batches, U[n..n+7] is padding, so the reads need no guard
            x[t] = (FULL || e < m) ? U[e] : 0.0;
            xn[t] = (O::NEXT && (FULL || e < m)) ? U[e + 1] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = e0 + t * LPR;
            if (e < m) {
                double a, b;
                O::term(x[t], xn[t], e, a, b);
                A[e] = a;
                if (O::TWO) B[e] = b;
            }
        }
    }
    lds_wave_fence();  // terms complete
    double sa, sb;
    if constexpr (NFIX != 0 && NFIX <= 256) {  // the number of terms, and with it numpy's plan, is known at compile time
        if constexpr (!O::NEXT)
            row_reduce_fixed<O::TWO, O::BMUL, LPR, NFIX>(A, B, l, sa, sb);
        else
            row_reduce_static<O::TWO, O::BMUL, NFIX - 1>(A, B, l, sa, sb);
        return O::finish(sa, sb, n);
    }
    // uniform: n > 128, leaves reduced in parallel by the row's 8-lane groups.  (For FULL rows the choice is known at
    // compile time, and dropping the other branch takes the two-stream PSO kernels from 106 to 81 VGPRs = 6 waves per
    // SIMD instead of 4 -- measured SLOWER at BASELINE config 3, 48.2 vs 45.0 us: profiles/r2_pso_c3_variants.txt.)
    if (plan.nleaf > 1)
        row_reduce_leaves<O::TWO, O::BMUL, LPR>(A, B, B + n, B + n + 24, leaf_cap(n), plan, l, sa, sb);
    else
        row_reduce2<O::TWO, O::BMUL>(A, B, B + n, plan, l, sa, sb);
    return O::finish(sa, sb, n);
}

// one (min f, first row) record per workgroup for the best-of-generation step
template <int LPR>
__device__ __forceinline__ void block_partial(double val, const RowIds<LPR> &id, double *sf, int64_t *si,
                                              double *__restrict__ part_f, int64_t *__restrict__ part_i) {
    if (id.l == 0) {
        sf[id.slot] = id.active ? val : __builtin_huge_val();
        si[id.slot] = id.active ? id.row : INT64_MAX;
    }
    __syncthreads();
    // wavefront 0 takes slot k in lane k (slots are in row order) and finds the record with one DPP minimum + ballot.
    // (One lane folding the slots in a 15-step compare-and-select chain cost 0.8 us at the end of every workgroup of
    // the metric shape -- 8.4 -> 7.6 us per generation.)
    if (threadIdx.x < kWave) {
        const int rows_in_block = (int)(blockDim.x >> 6) * RowIds<LPR>::RPW;
        const int k = (int)threadIdx.x;
        double bf = k < rows_in_block ? sf[k] : __builtin_huge_val();
        int64_t bi = k < rows_in_block ? si[k] : INT64_MAX;
        wave_argmin_ordered(bf, bi);
        if (k == 0) {
            part_f[id.block] = bf;
            part_i[id.block] = bi;
        }
    }
}

// ---- DE donor rows (shared by the synchronous and the sequential kernels)
constexpr int kMaxDonors = 5;

__host__ __device__ inline int donors_of(int strategy) {
    return strategy == SX_DE_RAND1BIN ? 3 : strategy == SX_DE_RAND2BIN ? 5 : strategy == SX_DE_BEST1BIN ? 2 : 4;
}

// Philox donors: k distinct rows != i, uniform without replacement.  Word 1+t of the
// donor calls gives r_t = mulhi(w, P-1-t); r_t is then shifted past the
// sorted exclusion list {i, d_0..d_{t-1}} (oracle/streams.py PhiloxStream.de_generation).
__device__ __forceinline__ void philox_donors(int64_t P, int k, int64_t i, uint32_t grow, uint32_t gen, uint32_t k0,
                                              uint32_t k1, int n, int64_t (&d)[kMaxDonors], int &irand) {
    // word 0 -> forced crossover index, word 1+t -> donor t; the second call only for 4- and 5-donor strategies
    const U4 a = philox4x32_10(0u, grow, gen, kPurposeDeDonor, k0, k1);
    U4 b = {0u, 0u, 0u, 0u};
    if (k > 3) b = philox4x32_10(1u, grow, gen, kPurposeDeDonor, k0, k1);
    const uint32_t w[5] = {a.y, a.z, a.w, b.x, b.y};
    uint32_t excl[kMaxDonors + 1];  // P < 2^31 (checked on the host): 32-bit index arithmetic
    excl[0] = (uint32_t)i;
    const uint32_t Pm1 = (uint32_t)(P - 1);
#pragma unroll
    for (int t = 0; t < kMaxDonors; ++t) {
        if (t < k) {
            uint32_t v = __umulhi(w[t], Pm1 - (uint32_t)t);
#pragma unroll
            for (int s = 0; s <= t; ++s) v += (v >= excl[s]) ? 1u : 0u;
            d[t] = (int64_t)v;
            uint32_t carry = v;  // insert v into the sorted list excl[0..t]
#pragma unroll
            for (int s = 0; s <= t; ++s) {
                if (carry < excl[s]) {
                    const uint32_t tmp = excl[s];
                    excl[s] = carry;
                    carry = tmp;
                }
            }
            excl[t + 1] = carry;
        } else {
            d[t] = 0;
        }
    }
    irand = (int)__umulhi(a.x, (uint32_t)n);
}

// mutant of one element (de/_strategy.py:1-38, the reference's association order; no FMA): g = best row,
// d0..d4 = donor rows
__device__ __forceinline__ double de_mutant(int strategy, double g, double d0, double d1, double d2, double d3,
                                            double d4, double F) {
    if (strategy == SX_DE_BEST1BIN) return g + F * (d0 - d1);
    if (strategy == SX_DE_RAND1BIN) return d0 + F * (d1 - d2);
    if (strategy == SX_DE_BEST2BIN) return g + F * (((d0 + d1) - d2) - d3);
    return d0 + F * (((d1 + d2) - d3) - d4);
}

// new velocity of one element (cpso/_cpso.py:326, left to right): w*v + c1*r1*(p - x) + c2*r2*(g - x)
__device__ __forceinline__ double pso_velocity(double w, double v, double c1, double r1, double p, double x, double c2,
                                               double r2, double g) {
    return (w * v + (c1 * r1) * (p - x)) + (c2 * r2) * (g - x);
}

// run `body(std::integral_constant<int, LPR>)` for the LPR that lanes_per_row(n) prescribes
#define SX_DISPATCH_LPR(n, CALL)            \
    switch (lanes_per_row(n)) {             \
        case 16: { constexpr int LPR = 16; CALL; } break; \
        case 32: { constexpr int LPR = 32; CALL; } break; \
        default: { constexpr int LPR = 64; CALL; } break; \
    }

}  // namespace sx
