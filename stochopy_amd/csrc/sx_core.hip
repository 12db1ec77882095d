// Core of the C ABI: error channel, summation plans, batched objective
// evaluation, argmin and the best-of-generation / termination kernel.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/factory/benchmark.py:14-156              the seven objectives
//   stochopy/optimize/_common.py:34-90                population wrapper fun(X) -> f
//   stochopy/optimize/_common.py:131-158              argmin + termination ladder
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"
#include "sx_wide.hpp"

namespace sx {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace sx

using namespace sx;

extern "C" int sx_abi_version(void) { return SX_ABI_VERSION; }
extern "C" const char *sx_last_error(void) { return g_error.c_str(); }
extern "C" int sx_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(sx_state);
        case 1: return (int)sizeof(sx_de_args);
        case 2: return (int)sizeof(sx_pso_args);
        case 3: return (int)sizeof(sx_xchg_args);
        case 4: return (int)sizeof(sx_cma_state);
        case 5: return (int)sizeof(sx_cma_args);
        case 6: return (int)sizeof(sx_vd_args);
    }
    return -1;
}
extern "C" int sx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return -1;
    }
    return n;
}

// ---------------------------------------------------------------------------
// numpy pairwise-sum plan (host)
// ---------------------------------------------------------------------------
namespace {
struct PlanBuilder {
    std::vector<int> end_block, merges;
    int blocks = 0;
    int depth = 0, max_depth = 0;
    void rec(int64_t m) {
        if (m <= 128) {
            blocks += (int)(m / 8);
            end_block.push_back(blocks);
            merges.push_back(0);
            ++depth;
            if (depth > max_depth) max_depth = depth;
            return;
        }
        int64_t h = m / 2;
        h -= h % 8;
        rec(h);
        rec(m - h);
        merges.back() += 1;
        --depth;
    }
};
}  // namespace

extern "C" int sx_sum_plan(int64_t m, int32_t *out, int cap) {
    if (m < 0 || cap < 4) return -1;
    if (m < 8) {
        out[0] = 0;
        out[1] = (int32_t)m;
        out[2] = 0;
        out[3] = 1;
        return 4;
    }
    PlanBuilder b;
    b.rec(m);
    const int nleaf = (int)b.end_block.size();
    const int need = 4 + 2 * nleaf;
    if (cap < need) return -need;
    out[0] = nleaf;
    out[1] = (int32_t)(m % 8);
    out[2] = (int32_t)(m / 8);
    out[3] = b.max_depth;
    for (int t = 0; t < nleaf; ++t) {
        out[4 + 2 * t] = b.end_block[t];
        out[5 + 2 * t] = b.merges[t];
    }
    return need;
}

extern "C" int64_t sx_fun_terms(int fun_id, int n) {
    if (fun_id == SX_FUN_ROSENBROCK) return n > 0 ? n - 1 : 0;
    return n;
}

extern "C" int64_t sx_num_partials(int64_t P, int n) { return (int64_t)row_geometry(P, n).blocks; }
namespace sx {
int g_wide_from = kWideFrom;
}
extern "C" int sx_rows_per_workgroup(int n) { return n > sx::wide_from() ? 1 : rows_per_block(n); }
extern "C" int sx_wide_from(void) { return sx::wide_from(); }
// n <= 0: back to the library's own threshold; otherwise clamped to [256, 4096] (what the wavefront-per-row kernels can serve).
// Returns the previous value.  Process-wide, not thread-safe: set it before a run's first call and restore it after its last
// (everything that depends on it -- record counts, geometries, which kernels run -- is read per call).
extern "C" int sx_set_wide_from(int n) {
    const int prev = sx::g_wide_from;
    sx::g_wide_from = n <= 0 ? kWideFrom : (n < 256 ? 256 : (n > kMaxDim ? kMaxDim : n));
    return prev;
}

namespace sx {
int make_plan_arg(int fun_id, int n, PlanArg *out) {
    if (n > kMaxDim) {  // (rows the wide kernels serve take their plan from device memory: sx_wide.hip; callers branch before this --
                        //  except sx_eval's eight-lanes-per-row form, which walks rows of up to kMaxDim elements with this plan)
        set_error("internal: a narrow-row kernel was asked for a row the wide kernels serve");
        return -1;
    }
    const int64_t m = sx_fun_terms(fun_id, n);
    std::vector<int32_t> buf(4 + 2 * (size_t)(m / 64 + 2));
    const int got = sx_sum_plan(m, buf.data(), (int)buf.size());
    if (got < 0 || buf[0] > kMaxLeaf || buf[3] > 12) {
        set_error("dimension too large for the kernel-argument summation plan");
        return -1;
    }
    out->nleaf = buf[0];
    out->tail = buf[1];
    out->mb = buf[2];
    out->depth = buf[3];
    std::vector<int> stack;  // slots (leaf indices) of the pending partial sums
    int nm = 0;
    for (int t = 0; t < buf[0]; ++t) {
        out->end[t] = buf[4 + 2 * t];
        out->merges[t] = buf[5 + 2 * t];
        stack.push_back(t);
        for (int k = 0; k < buf[5 + 2 * t]; ++k) {
            const int right = stack.back();
            stack.pop_back();
            out->mleft[nm] = stack.back();
            out->mright[nm] = right;
            ++nm;
        }
    }
    // work slots: consecutive leaves of <= 8 blocks each share one (sx_device.hpp, PlanArg)
    int ns = 0;
    for (int t = 0; t < buf[0];) {
        const int b0 = t > 0 ? out->end[t - 1] : 0, b1 = out->end[t];
        const bool pair = t + 1 < buf[0] && b1 - b0 <= 8 && out->end[t + 1] - b1 <= 8;
        out->sfirst[ns++] = t;
        t += pair ? 2 : 1;
    }
    if ((ns + 7) / 8 == (buf[0] + 7) / 8) {  // no pass saved (eight groups per pass): one leaf per slot, the plain form
        ns = buf[0];
        for (int t = 0; t < ns; ++t) out->sfirst[t] = t;
    }
    out->nslot = ns;
    out->sfirst[ns] = buf[0];
    return 0;
}
}  // namespace sx

// ---------------------------------------------------------------------------
// Batched evaluation kernel: one wavefront per individual.
// ---------------------------------------------------------------------------
// FULL: n is a whole number of 4-step batches and P a whole number of workgroups (no bounds test survives);
// NFIX: FULL with n == 4 * LPR exactly (64 / 128 / 256): the row length, and with it numpy's summation plan, is a
// compile-time constant (row_reduce_fixed / row_reduce_static, as in the one-batch DE / PSO kernels); 0 otherwise.
template <int FUN, int LPR, bool FULL = false, int NFIX = 0>
__global__ __launch_bounds__(kMaxWavesPerBlock *kWave) void eval_kernel(
    const double *__restrict__ X, int64_t P, int n_arg, int64_t ldx, const double *__restrict__ xm,
    const double *__restrict__ xstd, double *__restrict__ f, const PlanArg plan, double *__restrict__ part_f,
    int64_t *__restrict__ part_i, const int clip, const double *__restrict__ pen_v, double *__restrict__ pen_out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sf[kMaxRowsPerBlock];
    __shared__ int64_t si[kMaxRowsPerBlock];
    const int n = NFIX ? NFIX : n_arg;
    const RowIds<LPR> id(P);
    double *U = lds + id.slot * gen_row_stride(n);  // (n + 8 doubles up to 256 elements: launch_eval sizes the LDS to match)
    const double *xr = X + id.rowc * ldx;
    const bool affine = xm != nullptr;
    double pacc = 0.0;
    constexpr int kBatch = (NFIX && NFIX <= 256) ? 4 : 8;  // row loads of a lane in flight together (the kernel is a pure stream of rows)
    // (An explicit copy of this loop without the CMA-ES switches for plain calls was measured: no gain -- the compiler already
    // unswitches it here, unlike in the wide form of this kernel; profiles/r5_eval_mid_ab.txt.)
    for (int e0 = id.l; e0 < n; e0 += kBatch * LPR) {
        double xv[kBatch];
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
            const int e = e0 + t * LPR;
            xv[t] = (FULL || e < n) ? xr[e] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < kBatch; ++t) {
            const int e = e0 + t * LPR;
            if (!FULL && e >= n) continue;
            double v = xv[t];
            if (clip) {  // cmaes/_constraints.py:29-31 (clip to the standardised box), :79 (weighted squared excess)
                const double c = v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
                if (pen_v != nullptr) pacc += ((c - v) * (c - v)) * pen_v[e];
                v = c;
            }
            if (affine) v = v * xstd[e] + xm[e];  // cmaes/_cmaes.py:171 unstandardize
            U[e] = v;
        }
    }
    if (pen_out != nullptr) {
        pacc = row_sum<LPR>(pacc);
        if (id.active && id.l == 0) pen_out[id.row] = pacc;
    }
    const double val = row_objective<FUN, LPR, FULL, NFIX>(U, n, plan, id.l);
    if ((FULL || id.active) && id.l == 0) f[id.row] = val;
    if (part_f != nullptr) block_partial<LPR>(val, id, sf, si, part_f, part_i);
}

// Eight lanes per row (round 5).  What bounds one-batch rows at large P turned out to be instruction issue, not memory
// (profiles/r5_eval_stream_ab.txt: a resident form of the one-visit kernel -- a fixed grid of wavefronts walking the row blocks
// with the loads two visits ahead, 16-byte lane loads -- was built first and moved nothing: 0.49 -> 0.51 at n = 128; and
// Rosenbrock n=64 needs 222 us where Sphere n=64 needs 167): with 16 / 32 / 64 lanes per row every wavefront instruction of the accumulator chains serves 4 / 2 / 1 rows --
// the chains are sequential, so three quarters of the lanes idle through each of their 15 additions -- and the seven tail
// terms are formed by every lane.  Here a row belongs to EIGHT lanes, one per accumulator of numpy's sum, and a wavefront
// carries eight rows: lane (r, j) forms the terms of elements 8k + j of row r and adds them in order (16 terms and 15 additions
// per lane at n = 128, none of them idle), the tree is three DPP steps inside the 8-lane group, the tail terms are formed by
// lanes j = 0..6 (one term each) and added in order by a DPP scan along the group.  The rows reach LDS by 16-byte lane loads
// (one instruction = 1 KB of consecutive memory).  Row stride n + 8 doubles: the 32 lanes of a ds_read_b64 group (4 rows x 8
// accumulators) hit 32 different bank pairs.  Same terms, same additions in numpy's order: same bits.
template <int FUN, int OFF, int MM>
__device__ __forceinline__ void r8_sum(const double *U, int j, double &ra, double &rb) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    const double identB = BMUL ? 1.0 : 0.0;
    if constexpr (MM <= 128) {
        static_assert(MM >= 8, "shorter sums are plain loops");
        constexpr int BLK = MM / kGroup, TAIL = MM % kGroup;
        double chA = 0.0, chB = identB;
#pragma unroll
        for (int h0 = 0; h0 < BLK; h0 += 8) {  // eight blocks' reads in flight together
            double x[8], xn[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (h0 + t < BLK) {
                    x[t] = U[OFF + (h0 + t) * kGroup + j];
                    xn[t] = O::NEXT ? U[OFF + (h0 + t) * kGroup + j + 1] : 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (h0 + t < BLK) {
                    double a, b;
                    O::term(x[t], xn[t], OFF + (h0 + t) * kGroup + j, a, b);
                    if (h0 + t == 0) {
                        chA = a, chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        }
        ra = group_tree<false>(chA);
        rb = TWO ? group_tree<BMUL>(chB) : identB;
        if constexpr (TAIL > 0) {
            // tail term t by lane t; the running sum walks along the group (row_shr:1 -- lane j takes lane j-1's value), so
            // after step t lane t holds (((tree + a_0) + a_1) ... + a_t): the leaf's sum ends in lane TAIL-1
            const int e = OFF + BLK * kGroup + (j < TAIL ? j : 0);
            double ta, tb;
            O::term(U[e], O::NEXT ? U[e + 1] : 0.0, e, ta, tb);
            ra = ra + ta;
            if (TWO) rb = combine<BMUL>(rb, tb);
#pragma unroll
            for (int t = 1; t < TAIL; ++t) {
                const double pa = dpp_f64<0x111>(ra) + ta;
                ra = j >= t ? pa : ra;
                if (TWO) {
                    const double pb = combine<BMUL>(dpp_f64<0x111>(rb), tb);
                    rb = j >= t ? pb : rb;
                }
            }
        }
    } else {  // numpy's split: n/2 rounded down to a multiple of 8; only the right part can have a tail
        constexpr int N2 = (MM / 2) - ((MM / 2) % kGroup);
        double la, lb, qa, qb;
        r8_sum<FUN, OFF, N2>(U, j, la, lb);
        r8_sum<FUN, OFF + N2, MM - N2>(U, j, qa, qb);
        ra = la + qa;
        rb = TWO ? combine<BMUL>(lb, qb) : identB;
    }
}

template <int FUN, int NFIX>
__global__ __launch_bounds__(256) void eval_r8_kernel(const double *__restrict__ X, int64_t ldx, double *__restrict__ f) {
    using O = Obj<FUN>;
    constexpr int M = O::NEXT ? NFIX - 1 : NFIX, STRIDE = NFIX + 8, NLD = 8 * NFIX / 128;  // 16-byte lane loads per wavefront visit
    constexpr int TAIL = M % kGroup;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    double *W = lds + wave * 8 * STRIDE;
    const int64_t row0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 8;
    double2 v[NLD];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int idx = t * 128 + 2 * lane, r = idx / NFIX, c = idx % NFIX;
        v[t] = *reinterpret_cast<const double2 *>(X + (row0 + r) * ldx + c);
    }
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int idx = t * 128 + 2 * lane, r = idx / NFIX, c = idx % NFIX;
        *reinterpret_cast<double2 *>(W + r * STRIDE + c) = v[t];
    }
    lds_wave_fence();
    const int r = lane >> 3, j = lane & 7;
    double sa, sb;
    r8_sum<FUN, 0, M>(W + r * STRIDE, j, sa, sb);
    sa = 0.0 + sa;  // add.reduce starts from the identity
    sb = !O::TWO ? (O::BMUL ? 1.0 : 0.0) : (O::BMUL ? sb : 0.0 + sb);
    const double val = O::finish(sa, sb, NFIX);
    if (j == (TAIL > 0 ? TAIL - 1 : 0)) f[row0 + r] = val;
}

// The same mapping for one-batch rows of ANY length (n <= 256: at most three leaves of at most 16 blocks, PlanArg), straight from
// memory: lane (r, j) of a wavefront owns accumulator j of row r, so the elements it needs -- 8k + j -- are exactly the ones it
// loads (eight rows x 64 consecutive bytes per load instruction, consecutive blocks in consecutive instructions: every 128-byte
// line is fetched once and hit once); nothing is staged.  The neighbour of a Rosenbrock-like term is a second load of the
// same lines (taking it from the next lane by a DPP move, with lane 7 alone loading, was measured: 0.58 -> 0.36 of the HBM peak --
// the eight-lane loads cost more than the full ones they replace; profiles/r5_eval_r8_rt.txt, part 5).  Rows off the compile-time grid (n = 100, 200, ...) otherwise take the 16 / 32 / 64-lanes-per-row kernel:
// Rosenbrock n = 100 / 130 / 200 / 250 at large P: 0.40 / 0.20 / 0.28 / 0.34 of the HBM peak -> 0.65 / 0.62 / 0.64 / 0.61; Ackley 0.20 / 0.12
// / 0.20 / 0.22 -> 0.47 / 0.48 / 0.54 / 0.52 (profiles/r5_eval_r8_rt.txt).
template <int FUN>
__global__ __launch_bounds__(256) void eval_r8_rt_kernel(const double *__restrict__ X, int64_t P, int n, int64_t ldx,
                                                         double *__restrict__ f, const PlanArg plan) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    const double identB = BMUL ? 1.0 : 0.0;
    const int lane = (int)(threadIdx.x & 63), j = lane & 7;
    const int64_t row = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool live = row < P;
    const double *__restrict__ xr = X + (live ? row : P - 1) * ldx;
    const int nbt = plan.mb, tail = plan.tail, nleaf = plan.nleaf;
    double sA[3] = {0.0, 0.0, 0.0}, sB[3] = {identB, identB, identB};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t >= nleaf) break;  // (uniform)
        const int b0 = t > 0 ? plan.end[t - 1] : 0, cnt = plan.end[t] - b0;
        double chA = 0.0, chB = identB;
        // (round 6: objectives without a neighbour term take the whole leaf's 16 loads in one batch -- twice the bytes in
        //  flight per wave in front of the 16 cosines; SX_EVAL_RT_HB=8 restores the batches of eight for an A/B)
#ifndef SX_EVAL_RT_HB
#define SX_EVAL_RT_HB 16
#endif
        constexpr int HB = O::NEXT ? 8 : SX_EVAL_RT_HB;
#pragma unroll
        for (int h0 = 0; h0 < kLeafBlocks; h0 += HB) {
            double x[HB], xn[O::NEXT ? HB : 1];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool in = h0 + u < cnt;
                const int e = (b0 + h0 + u) * kGroup + j;
                x[u] = in ? xr[e] : 0.0;
                if constexpr (O::NEXT) xn[u] = in ? xr[e + 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool in = h0 + u < cnt;
                const int e = (b0 + h0 + u) * kGroup + j;
                double a, b;
                if constexpr (light_objective<FUN>()) {  // a select per term; the cosine objectives branch (registers)
                    O::term(x[u], O::NEXT ? xn[O::NEXT ? u : 0] : 0.0, e, a, b);
                    if (h0 + u == 0) {
                        chA = in ? a : 0.0;
                        chB = in ? b : identB;
                    } else {
                        chA = in ? chA + a : chA;
                        if (TWO) chB = in ? combine<BMUL>(chB, b) : chB;
                    }
                } else if (in) {
                    O::term(x[u], O::NEXT ? xn[O::NEXT ? u : 0] : 0.0, e, a, b);
                    if (h0 + u == 0) {
                        chA = a;
                        chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        }
        sA[t] = group_tree<false>(chA);
        sB[t] = TWO ? group_tree<BMUL>(chB) : identB;
    }
    // the last leaf takes the tail terms one by one: term t by lane t, the running sum walks along the group (r8_sum's scan)
    double la = nleaf == 0 ? 0.0 : nleaf == 1 ? sA[0] : nleaf == 2 ? sA[1] : sA[2];
    double lb = nleaf == 0 ? identB : nleaf == 1 ? sB[0] : nleaf == 2 ? sB[1] : sB[2];
    if (tail > 0) {  // (uniform)
        const int e = kGroup * nbt + (j < tail ? j : 0);
        double ta, tb;
        O::term(xr[e], O::NEXT ? xr[e + 1] : 0.0, e, ta, tb);
        la = la + ta;
        if (TWO) lb = combine<BMUL>(lb, tb);
#pragma unroll
        for (int t = 1; t < kGroup - 1; ++t) {
            const double pa = dpp_f64<0x111>(la) + ta;
            const bool on = t < tail && j >= t;
            la = on ? pa : la;
            if (TWO) {
                const double pb = combine<BMUL>(dpp_f64<0x111>(lb), tb);
                lb = on ? pb : lb;
            }
        }
        const int src = (lane & ~(kGroup - 1)) + tail - 1;  // the sum ends in lane tail - 1 of the group
        la = __shfl(la, src, kWave);
        if (TWO) lb = __shfl(lb, src, kWave);
    }
    // merges in recursion order: S0 + S1, or S0 + (S1 + S2)
    double sa, sb;
    if (nleaf > 2) {
        sa = sA[0] + (sA[1] + la);
        sb = TWO ? combine<BMUL>(sB[0], combine<BMUL>(sB[1], lb)) : identB;
    } else if (nleaf > 1) {
        sa = sA[0] + la;
        sb = TWO ? combine<BMUL>(sB[0], lb) : identB;
    } else {
        sa = la, sb = lb;
    }
    sa = 0.0 + sa;  // add.reduce starts from the identity
    sb = !TWO ? identB : (BMUL ? sb : 0.0 + sb);
    const double val = O::finish(sa, sb, n);
    if (live && j == 0) f[row] = val;
}
// ... and for rows of 257 ... kMaxDim = 4096 elements of any length: the group walks its row's leaves one after the other (numpy's plan
// has up to 17 of them at n = 2048), leaves its sums in LDS (2 nleaf doubles per row) and lane 0 of the group performs the
// recursion's combines in order.  Against the wavefront-per-row kernel (row staged in LDS, one leaf per 8-lane group, plan
// walked with scalar loads and selects): profiles/r5_eval_r8_rt.txt, part 3.
template <int FUN>
__global__ __launch_bounds__(256) void eval_r8_long_kernel(const double *__restrict__ X, int64_t P, int n, int64_t ldx,
                                                           double *__restrict__ f, const PlanArg plan) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    const double identB = BMUL ? 1.0 : 0.0;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = (int)(threadIdx.x & 63), j = lane & 7, grp = (int)(threadIdx.x >> 3);
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 3) + grp;
    const bool live = row < P;
    const double *__restrict__ xr = X + (live ? row : P - 1) * ldx;
    const int nbt = plan.mb, tail = plan.tail, nleaf = plan.nleaf;
    double *LA = lds + (size_t)grp * 2 * nleaf, *LB = LA + nleaf;
    for (int t = 0; t < nleaf; ++t) {  // (uniform)
        const int b0 = t > 0 ? plan.end[t - 1] : 0, cnt = plan.end[t] - b0;
        double chA = 0.0, chB = identB;
        // (round 6: objectives without a neighbour term take the whole leaf's 16 loads in one batch -- twice the bytes in
        //  flight per wave in front of the 16 cosines; SX_EVAL_RT_HB=8 restores the batches of eight for an A/B)
#ifndef SX_EVAL_RT_HB
#define SX_EVAL_RT_HB 16
#endif
        constexpr int HB = O::NEXT ? 8 : SX_EVAL_RT_HB;
#pragma unroll
        for (int h0 = 0; h0 < kLeafBlocks; h0 += HB) {
            double x[HB], xn[O::NEXT ? HB : 1];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool in = h0 + u < cnt;
                const int e = (b0 + h0 + u) * kGroup + j;
                x[u] = in ? xr[e] : 0.0;
                if constexpr (O::NEXT) xn[u] = in ? xr[e + 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const bool in = h0 + u < cnt;
                const int e = (b0 + h0 + u) * kGroup + j;
                double a, b;
                if constexpr (light_objective<FUN>()) {
                    O::term(x[u], xn[u], e, a, b);
                    if (h0 + u == 0) {
                        chA = in ? a : 0.0;
                        chB = in ? b : identB;
                    } else {
                        chA = in ? chA + a : chA;
                        if (TWO) chB = in ? combine<BMUL>(chB, b) : chB;
                    }
                } else if (in) {
                    O::term(x[u], xn[u], e, a, b);
                    if (h0 + u == 0) {
                        chA = a;
                        chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        }
        double la = group_tree<false>(chA), lb = TWO ? group_tree<BMUL>(chB) : identB;
        if (t == nleaf - 1 && tail > 0) {  // (uniform) the tail terms: term k by lane k, the running sum walks along the group
            const int e = kGroup * nbt + (j < tail ? j : 0);
            double ta, tb;
            O::term(xr[e], O::NEXT ? xr[e + 1] : 0.0, e, ta, tb);
            la = la + ta;
            if (TWO) lb = combine<BMUL>(lb, tb);
#pragma unroll
            for (int k = 1; k < kGroup - 1; ++k) {
                const double pa = dpp_f64<0x111>(la) + ta;
                const bool on = k < tail && j >= k;
                la = on ? pa : la;
                if (TWO) {
                    const double pb = combine<BMUL>(dpp_f64<0x111>(lb), tb);
                    lb = on ? pb : lb;
                }
            }
            if (j == tail - 1) {
                LA[t] = la;
                if (TWO) LB[t] = lb;
            }
        } else if (j == 0) {
            LA[t] = la;
            if (TWO) LB[t] = lb;
        }
    }
    lds_wave_fence();
    if (j == 0) {  // the recursion's combines in order (slots = leaf indices; the total ends in slot 0)
        for (int mm = 0; mm + 1 < nleaf; ++mm) {
            const int left = plan.mleft[mm], right = plan.mright[mm];
            LA[left] = LA[left] + LA[right];
            if (TWO) LB[left] = combine<BMUL>(LB[left], LB[right]);
        }
        double sa = 0.0 + LA[0];  // add.reduce starts from the identity
        double sb = !TWO ? identB : (BMUL ? LB[0] : 0.0 + LB[0]);
        if (live) f[row] = O::finish(sa, sb, n);
    }
}
// mode 2 (measurement): also the cheap objectives at n = 512 / 1024 / 2048 and beyond 3584 elements
static int eval_r8_long_mode() {
    static const int mode = getenv("SX_EVAL_R8LONG") ? atoi(getenv("SX_EVAL_R8LONG")) : 1;
    return mode;
}
// From which population size on (profiles/r5_eval_small_p.txt: a row is a serial walk of n / 8 blocks here, so a small population
// is faster one wavefront per row): off the compile-time grid 8192 rows (a cosine per term: 16 384); where the alternative is a
// compile-time plan or the workgroup per row (n = 512 / 1024 / 2048, n > 2048), measured at large P only: 32 768.
static int64_t eval_r8_min_rows(int64_t dflt) {
    static const int64_t forced = getenv("SX_EVAL_R8_MIN") ? atoll(getenv("SX_EVAL_R8_MIN")) : -1;
    return forced >= 0 ? forced : dflt;
}
static bool eval_r8_long_ok(int64_t P, int n, const double *xm, const double *part_f, int clip, int nleaf, bool cheap) {
    const bool has_rival = n > sx::wide_from() || n == 512 || n == 1024 || n == 2048;
    return eval_r8_long_mode() != 0 && n > 256 && n <= kMaxDim && nleaf >= 1 && nleaf <= kMaxLeaf && xm == nullptr &&
           part_f == nullptr && clip == 0 && P >= eval_r8_min_rows(has_rival ? 32768 : cheap ? 8192 : 16384);
}

// mode 1: rows off the compile-time grid; 2: every one-batch row (measurement: against eval_r8_kernel at n = 64 / 128 / 256)
static int eval_r8_rt_mode() {
    static const int mode = getenv("SX_EVAL_R8RT") ? atoi(getenv("SX_EVAL_R8RT")) : 1;
    return mode;
}
static bool eval_r8_rt_ok(int64_t P, int n, const double *xm, const double *part_f, int clip) {
    const bool on_grid = n == 64 || n == 128 || n == 256;
    return eval_r8_rt_mode() != 0 && n >= 16 && n <= 256 && xm == nullptr && part_f == nullptr && clip == 0 &&
           P >= eval_r8_min_rows(on_grid ? 32768 : 8192);
}

#ifndef SX_EVAL_HEAVY_STATIC
#define SX_EVAL_HEAVY_STATIC 1  // (0: objectives with a cosine per term keep the run-time plan on long rows)
#endif
static int device_cus() {
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return cus;
}
// eight lanes per row: plain evaluations of whole wavefront loads of 16-byte aligned one-batch rows, from the population size at
// which the chip is full either way (below it the one-visit kernel's 16 rows per workgroup spread a small population wider)
static bool eval_r8_ok(const double *X, int64_t P, int n, int64_t ldx, const double *xm, const double *part_f) {
    static const int mode = getenv("SX_EVAL_R8") ? atoi(getenv("SX_EVAL_R8")) : 1;
    static const int64_t min_rows = getenv("SX_EVAL_R8_MIN") ? atoll(getenv("SX_EVAL_R8_MIN")) : 32768;
    return mode != 0 && (n == 64 || n == 128 || n == 256) && xm == nullptr && part_f == nullptr && (ldx & 1) == 0 &&
           ((uintptr_t)X & 15) == 0 && P % 32 == 0 && P >= min_rows;
}
template <int FUN>
static int launch_eval(const double *X, int64_t P, int n, int64_t ldx, const double *xm, const double *xstd, double *f,
                       const PlanArg &plan, double *part_f, int64_t *part_i, hipStream_t s, int clip = 0,
                       const double *pen_v = nullptr, double *pen_out = nullptr) {
    const Geometry g = row_geometry(P, n);
    // whole batches and whole workgroups: the guard-free form; one batch per row on top: the compile-time plan
    // (plain evaluation only -- the clip / penalty variants keep the general kernel, to keep the build small)
    const int lpr = lanes_per_row(n);
    const bool full = clip == 0 && n % (8 * lpr) == 0 && P % rows_per_block(n) == 0;
    // (the compile-time row lengths -- one-batch rows, n = 512 / 1024 / 2048 -- for the four hot objectives only: hot_objective)
    constexpr bool kHot = FUN == SX_FUN_ACKLEY || FUN == SX_FUN_RASTRIGIN || FUN == SX_FUN_ROSENBROCK || FUN == SX_FUN_SPHERE;
    const bool fix = kHot && clip == 0 && n == 4 * lpr && P % rows_per_block(n) == 0;
    size_t lds = (size_t)rows_per_block(n) * gen_row_stride(n) * sizeof(double);  // (n + 8 doubles per row up to 256 elements)
    // (objectives with a cosine per term keep the run-time plan: eight inlined cosines side by side need more registers than
    //  a wavefront slot has, and their arithmetic, not the plan, is their time)
    constexpr bool kLight = light_objective<FUN>();
#define SX_EVAL_GO(...)                                                                                              \
    hipLaunchKernelGGL((eval_kernel<FUN, __VA_ARGS__>), dim3(g.blocks), dim3(g.threads), lds, s, X, P, n, ldx, xm, xstd, f, \
                       plan, part_f, part_i, clip, pen_v, pen_out)
    const bool grid_long = kHot && clip == 0 && (n == 512 || n == 1024 || n == 2048) && P % rows_per_block(n) == 0;
    if (eval_r8_long_ok(P, n, xm, part_f, clip, plan.nleaf, kLight) && (eval_r8_long_mode() == 2 || !(kLight && grid_long))) {
        // many long rows: eight lanes per row, straight from memory.  Not the cheap objectives at n = 512 / 1024 / 2048: their
        // compile-time plan below stays ahead (Rosenbrock 0.61 / 0.76 / 0.66 of the HBM peak against 0.58 / 0.54 / 0.55);
        // the cosine objectives gain there too (Ackley 0.30 / 0.47 / 0.35 -> 0.49 / 0.50 / 0.48): profiles/r5_eval_r8_rt.txt
        hipLaunchKernelGGL((eval_r8_long_kernel<FUN>), dim3((unsigned)((P + 31) / 32)), dim3(256),
                           (size_t)32 * 2 * plan.nleaf * sizeof(double), s, X, P, n, ldx, f, plan);
    } else if ((kLight || SX_EVAL_HEAVY_STATIC) && grid_long) {
        // long rows of a compile-time length: numpy's plan as constants (row_reduce_long).  Rosenbrock n = 1024: 0.54 -> 0.84
        // of the HBM peak, n = 512: 0.44 -> 0.76, n = 2048: 0.43 -> 0.70 (profiles/r4_eval_long_rows.txt; a resident,
        // software-pipelined form of the same kernel stayed at 0.72 and was dropped)
        if constexpr (kHot) {
            switch (n) {
                case 512: SX_EVAL_GO(64, true, 512); break;
                case 1024: SX_EVAL_GO(64, true, 1024); break;
                default: SX_EVAL_GO(64, true, 2048); break;
            }
        }
    } else if (fix && (eval_r8_rt_mode() == 2 || (n == 256 && !kLight)) && eval_r8_rt_ok(P, n, xm, part_f, clip)) {
        // rows of 256 elements with a cosine per term: the run-time form (no staging, 16 terms at a time) is the faster one --
        // Ackley 0.45 -> 0.57 of the HBM peak, Rastrigin 0.46 -> 0.59; at n = 64 / 128 and for the cheap objectives the
        // compile-time form below stays ahead (Rosenbrock 0.75-0.78 against 0.48-0.63): profiles/r5_eval_r8_rt.txt
        hipLaunchKernelGGL((eval_r8_rt_kernel<FUN>), dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, X, P, n, ldx, f, plan);
    } else if (fix && FUN != SX_FUN_SPHERE && eval_r8_ok(X, P, n, ldx, xm, part_f)) {
        // (Sphere -- one multiplication per element, no second stream -- is the one objective the one-visit kernel streams
        //  faster: 0.81 / 0.84 against 0.78 / 0.77 of the HBM peak at n = 64 / 256, profiles/r5_eval_r8_ab.txt)
        // plain evaluation of many one-batch rows: eight lanes per row, eight rows per wavefront
        const unsigned blocks8 = (unsigned)(P / 32);
        if constexpr (kHot) {
            switch (n) {
                case 64: hipLaunchKernelGGL((eval_r8_kernel<FUN, 64>), dim3(blocks8), dim3(256), 32 * (64 + 8) * sizeof(double), s, X, ldx, f); break;
                case 128: hipLaunchKernelGGL((eval_r8_kernel<FUN, 128>), dim3(blocks8), dim3(256), 32 * (128 + 8) * sizeof(double), s, X, ldx, f); break;
                default: hipLaunchKernelGGL((eval_r8_kernel<FUN, 256>), dim3(blocks8), dim3(256), 32 * (256 + 8) * sizeof(double), s, X, ldx, f); break;
            }
        }
    } else if (!fix && eval_r8_rt_ok(P, n, xm, part_f, clip) && plan.nleaf <= 3) {
        // many one-batch rows of a length off the compile-time grid: eight lanes per row, straight from memory
        hipLaunchKernelGGL((eval_r8_rt_kernel<FUN>), dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, X, P, n, ldx, f, plan);
    } else if (fix) {
        // the register-chain objective reads the staged vector only: n + 8 doubles per row, not the term arrays' 3n + ...
        if constexpr (kHot) {
            switch (lpr) {
                case 16: SX_EVAL_GO(16, true, 64); break;
                case 32: SX_EVAL_GO(32, true, 128); break;
                default: SX_EVAL_GO(64, true, 256); break;
            }
        }
    } else if (full) {
        SX_DISPATCH_LPR(n, SX_EVAL_GO(LPR, true, 0))
    } else {
        SX_DISPATCH_LPR(n, SX_EVAL_GO(LPR, false, 0))
    }
#undef SX_EVAL_GO
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_eval(int fun_id, const double *X, int64_t P, int n, int64_t ldx, const double *xm,
                       const double *xstd, double *f, double *part_f, int64_t *part_i, void *stream) {
    SX_REQUIRE(X && f, "sx_eval: null pointer");
    SX_REQUIRE(P >= 1 && n >= 1 && ldx >= n, "sx_eval: bad shape");
    SX_REQUIRE(fun_id >= 0 && fun_id < SX_FUN_COUNT, "sx_eval: unknown fun_id");
    SX_REQUIRE((xm == nullptr) == (xstd == nullptr), "sx_eval: xm and xstd must be given together");
    SX_REQUIRE((part_f == nullptr) == (part_i == nullptr), "sx_eval: part_f and part_i must be given together");
    hipStream_t s = (hipStream_t)stream;
    // Large populations of rows of 2049 ... 4096 elements: the eight-lanes-per-row kernel (launch_eval) instead of one workgroup
    // per row -- Rosenbrock n = 2049 / 3000 0.45 / 0.49 -> 0.56 / 0.54 of the HBM peak, Ackley n = 2049 / 4096 0.28 / 0.43 ->
    // 0.50 / 0.48; the cheap objectives from ~3600 elements on stay with the workgroup per row (n = 4096: 0.59-0.74 against
    // 0.56-0.63: rows a power of two apart meet in the same memory channels).  profiles/r5_eval_r8_rt.txt, part 4.
    const bool cheap = fun_id == SX_FUN_ROSENBROCK || fun_id == SX_FUN_SPHERE || fun_id == SX_FUN_QUARTIC ||
                       fun_id == SX_FUN_STYBLINSKI_TANG;  // (light_objective<FUN>())
    const bool long_rows = n <= kMaxDim && eval_r8_long_ok(P, n, xm, part_f, 0, 1, cheap) &&
                           (eval_r8_long_mode() == 2 || !cheap || n <= 3584);
    if (is_wide(n) && !long_rows) return wide_eval(fun_id, X, P, n, ldx, xm, xstd, f, part_f, part_i, 0, nullptr, nullptr, s);
    if (is_wide(n) && n <= kWideMaxDim)
        if (int rc = wide_warm_plan(fun_id, n, s)) return rc;  // (the generations that follow are wide launches)
    PlanArg plan;
    if (make_plan_arg(fun_id, n, &plan)) return -1;
    switch (fun_id) {
#define SX_CASE(ID) \
    case ID:        \
        return launch_eval<ID>(X, P, n, ldx, xm, xstd, f, plan, part_f, part_i, s);
        SX_CASE(SX_FUN_ACKLEY)
        SX_CASE(SX_FUN_GRIEWANK)
        SX_CASE(SX_FUN_QUARTIC)
        SX_CASE(SX_FUN_RASTRIGIN)
        SX_CASE(SX_FUN_ROSENBROCK)
        SX_CASE(SX_FUN_SPHERE)
        SX_CASE(SX_FUN_STYBLINSKI_TANG)
#undef SX_CASE
    }
    return -1;
}

// CMA-ES "Penalize" boundary handling (cmaes/_constraints.py:4-82), device part: candidates are clipped to the
// standardised box [-1, 1]^n before the objective (f_raw), and pen[i] = sum_j (clip(x_ij) - x_ij)^2 * v[j]
// (v = bnd_weights / bnd_scale from the host; NULL = no penalty term wanted).  One kernel.
extern "C" int sx_cmaes_eval_penalized(int fun_id, const double *X, int64_t P, int n, const double *xm,
                                       const double *xstd, const double *v, double *f_raw, double *pen, void *stream) {
    SX_REQUIRE(X && f_raw && xm && xstd, "sx_cmaes_eval_penalized: null pointer");
    SX_REQUIRE(P >= 1 && n >= 1, "sx_cmaes_eval_penalized: bad shape");
    SX_REQUIRE(fun_id >= 0 && fun_id < SX_FUN_COUNT, "sx_cmaes_eval_penalized: unknown fun_id");
    SX_REQUIRE((v == nullptr) == (pen == nullptr), "sx_cmaes_eval_penalized: v and pen must be given together");
    hipStream_t s = (hipStream_t)stream;
    if (is_wide(n)) return wide_eval(fun_id, X, P, n, n, xm, xstd, f_raw, nullptr, nullptr, 1, v, pen, s);
    PlanArg plan;
    if (make_plan_arg(fun_id, n, &plan)) return -1;
    switch (fun_id) {
#define SX_CASE(ID) \
    case ID:        \
        return launch_eval<ID>(X, P, n, n, xm, xstd, f_raw, plan, nullptr, nullptr, s, 1, v, pen);
        SX_CASE(SX_FUN_ACKLEY)
        SX_CASE(SX_FUN_GRIEWANK)
        SX_CASE(SX_FUN_QUARTIC)
        SX_CASE(SX_FUN_RASTRIGIN)
        SX_CASE(SX_FUN_ROSENBROCK)
        SX_CASE(SX_FUN_SPHERE)
        SX_CASE(SX_FUN_STYBLINSKI_TANG)
#undef SX_CASE
    }
    return -1;
}

// ---------------------------------------------------------------------------
// argmin: stage 1 (grid) -> partials, stage 2 (one workgroup) -> result
// ---------------------------------------------------------------------------
constexpr int kFinalThreads = 256;

__device__ __forceinline__ void block_argmin(double &bf, int64_t &bi, double *sf, int64_t *si) {
    wave_argmin_all(bf, bi);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sf[w] = bf;
        si[w] = bi;
    }
    __syncthreads();
    bf = sf[0];
    bi = si[0];
    for (int k = 1; k < kFinalThreads / kWave; ++k) argmin_combine(bf, bi, sf[k], si[k]);
    __syncthreads();
}

__global__ __launch_bounds__(kFinalThreads) void argmin_stage1(const double *__restrict__ f, int64_t P,
                                                               double *__restrict__ part_f,
                                                               int64_t *__restrict__ part_i) {
    __shared__ double sf[kFinalThreads / kWave];
    __shared__ int64_t si[kFinalThreads / kWave];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    for (int64_t k = (int64_t)blockIdx.x * kFinalThreads + threadIdx.x; k < P; k += (int64_t)gridDim.x * kFinalThreads)
        argmin_combine(bf, bi, f[k], k);
    block_argmin(bf, bi, sf, si);
    if (threadIdx.x == 0) {
        part_f[blockIdx.x] = bf;
        part_i[blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(kFinalThreads) void argmin_stage2(const double *__restrict__ part_f,
                                                               const int64_t *__restrict__ part_i, int64_t npart,
                                                               int64_t *__restrict__ out_idx,
                                                               double *__restrict__ out_val) {
    __shared__ double sf[kFinalThreads / kWave];
    __shared__ int64_t si[kFinalThreads / kWave];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    for (int64_t k = threadIdx.x; k < npart; k += kFinalThreads) argmin_combine(bf, bi, part_f[k], part_i[k]);
    block_argmin(bf, bi, sf, si);
    if (threadIdx.x == 0) {
        *out_idx = bi;
        *out_val = bf;
    }
}

extern "C" int sx_argmin(const double *f, int64_t P, double *ws_f, int64_t *ws_i, int64_t ws_len, int64_t *out_idx,
                         double *out_val, void *stream) {
    SX_REQUIRE(f && ws_f && ws_i && out_idx && out_val && P >= 1 && ws_len >= 1, "sx_argmin: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int64_t nblk = (P + kFinalThreads - 1) / kFinalThreads;
    if (nblk > ws_len) nblk = ws_len;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(argmin_stage1, dim3((unsigned)nblk), dim3(kFinalThreads), 0, s, f, P, ws_f, ws_i);
    SX_LAUNCH_CHECK();
    hipLaunchKernelGGL(argmin_stage2, dim3(1), dim3(kFinalThreads), 0, s, ws_f, ws_i, nblk, out_idx, out_val);
    SX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Best-of-generation + termination (stochopy/optimize/_common.py:131-158)
// ---------------------------------------------------------------------------
// best of the per-workgroup records, 8 records per thread and trip so their loads overlap
template <int kScan>
__device__ __forceinline__ void scan_records_n(const double *__restrict__ part_f, const int64_t *__restrict__ part_i,
                                               int64_t npart, double &bf, int64_t &bi) {
    for (int64_t k0 = threadIdx.x; k0 < npart; k0 += (int64_t)kFinalThreads * kScan) {
        double f[kScan];
        int64_t i[kScan];
#pragma unroll
        for (int u = 0; u < kScan; ++u) {
            const int64_t k = k0 + (int64_t)u * kFinalThreads;
            f[u] = k < npart ? part_f[k] : __builtin_huge_val();
            i[u] = k < npart ? part_i[k] : INT64_MAX;
        }
        // this trip's minimum (a tree), then the first record that holds it (k grows with u), then one lexicographic
        // step into the running pair: a chain of 8 compare-and-select steps otherwise
        double m = f[0];
#pragma unroll
        for (int u = 1; u < kScan; ++u) m = fmin(m, f[u]);  // (the minimum is the same in any order)
        int64_t first = i[0];  // (all NaN: the first record, as a sequential scan)
#pragma unroll
        for (int u = kScan - 1; u >= 0; --u)
            if (f[u] == m) first = i[u];
        argmin_combine(bf, bi, m, first);
    }
}
// 8 records per thread and trip; all of config 5 on one GPU (131 072 rows = 16 384 records, 64 per thread) 32: two trips
// instead of eight dependent ones
__device__ __forceinline__ void scan_records(const double *__restrict__ part_f, const int64_t *__restrict__ part_i,
                                             int64_t npart, double &bf, int64_t &bi) {
    if (npart > (int64_t)kFinalThreads * 16)
        scan_records_n<32>(part_f, part_i, npart, bf, bi);
    else
        scan_records_n<8>(part_f, part_i, npart, bf, bi);
}

// dst[e] = src[e] / sum of (a[e] - b[e])^2 over the elements of this thread (e = thread, thread + 256, ...: in that order), eight
// loads in flight per thread: the one-workgroup steps of the sharded path walk whole rows (up to 262 144 elements) with these
__device__ __forceinline__ void final_copy(double *__restrict__ dst, const double *__restrict__ src, int n) {
    for (int eb = threadIdx.x; eb < n; eb += 8 * kFinalThreads) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = eb + u * kFinalThreads < n ? src[eb + u * kFinalThreads] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (eb + u * kFinalThreads < n) dst[eb + u * kFinalThreads] = v[u];
    }
}
__device__ __forceinline__ double final_dist2(const double *__restrict__ a, const double *__restrict__ b, int n) {
    double acc = 0.0;
    for (int eb = threadIdx.x; eb < n; eb += 8 * kFinalThreads) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = eb + u * kFinalThreads < n;
            x[u] = in ? a[eb + u * kFinalThreads] : 0.0;
            y[u] = in ? b[eb + u * kFinalThreads] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (eb + u * kFinalThreads < n) {
                const double d = x[u] - y[u];
                acc += d * d;
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(kFinalThreads) void select_finalize_kernel(
    const double *__restrict__ part_f, const int64_t *__restrict__ part_i, int64_t npart,
    const double *__restrict__ rows0, const double *__restrict__ rows1, int64_t ld, int n,
    double *__restrict__ gbest, sx_state *__restrict__ state, int maxiter, double xtol, double ftol) {
    __shared__ double sf[kFinalThreads / kWave];
    __shared__ int64_t si[kFinalThreads / kWave];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    scan_records(part_f, part_i, npart, bf, bi);  // the loads do not depend on the state word: issue them first
    if (state->done) return;
    const int64_t it = state->it + 1;  // the generation being finalised
    block_argmin(bf, bi, sf, si);

    // generation g lives in rows[g & 1] (double-buffered populations); rows0 == rows1 for in-place state
    const double *src = ((it & 1) ? rows1 : rows0) + bi * ld;
    // dx = ||xbest_prev - x[k]||_2 (np.linalg.norm, _common.py:135); all loads of a thread in flight together
    constexpr int kPer = (kMaxDim + kFinalThreads - 1) / kFinalThreads;
    double gv[kPer], sv[kPer];
    double acc = 0.0;
    if (n <= kMaxDim) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = threadIdx.x + u * kFinalThreads;
            gv[u] = e < n ? gbest[e] : 0.0;
            sv[u] = e < n ? src[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (threadIdx.x + u * kFinalThreads < n) {
                const double d = gv[u] - sv[u];
                acc += d * d;
            }
        }
    } else {  // (longer rows take the three-launch form below: sx_select_finalize / add_finalize_node)
        acc = final_dist2(gbest, src, n);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) sf[threadIdx.x >> 6] = acc;
    __syncthreads();
    double ss = 0.0;
    for (int k = 0; k < kFinalThreads / kWave; ++k) ss += sf[k];
    const double dx = sqrt(ss);
    if (n <= kMaxDim) {
#pragma unroll
        for (int u = 0; u < kPer; ++u)
            if (threadIdx.x + u * kFinalThreads < n) gbest[threadIdx.x + u * kFinalThreads] = sv[u];
    } else {
        final_copy(gbest, src, n);
    }
    if (threadIdx.x == 0) {
        int status = SX_STATUS_NONE;
        if (dx <= xtol && bf <= ftol)
            status = 0;
        else if (bf <= ftol)
            status = 1;
        else if (it >= maxiter)
            status = -1;
        state->it = it;
        state->gbidx = bi;
        state->gfit = bf;
        state->dx = dx;
        state->status = status;
        state->done = status != SX_STATUS_NONE;
    }
}

// The same step for rows of more than kMaxDim elements, in three launches: one workgroup walking a row of 65 536 elements twice
// (the step of the best, then the copy) took 160 us per generation -- a third of a DE generation at that length
// (profiles/r5_wide_finalize.txt: select_finalize_kernel 47.6 us on average over the wide benchmark).  (1) the best record ->
// state->gbidx / gfit; (2) up to 64 workgroups: a slice of the row each -- its share of |gbest - x[best]|^2 (a fixed order: per
// thread, the wavefront's tree, the workgroup's four sums) into the first slots of the record buffer, which (1) has consumed,
// and the copy; (3) the shares added in order, the termination rules, the state.
constexpr int kWideFinalBlocks = 64;
__global__ __launch_bounds__(kFinalThreads) void wide_finalize_best_kernel(const double *__restrict__ part_f,
                                                                           const int64_t *__restrict__ part_i, int64_t npart,
                                                                           sx_state *__restrict__ state) {
    __shared__ double sf[kFinalThreads / kWave];
    __shared__ int64_t si[kFinalThreads / kWave];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    scan_records(part_f, part_i, npart, bf, bi);
    if (state->done) return;
    block_argmin(bf, bi, sf, si);
    if (threadIdx.x == 0) {
        state->gbidx = bi;
        state->gfit = bf;
    }
}
__global__ __launch_bounds__(kFinalThreads) void wide_finalize_row_kernel(const double *__restrict__ rows0,
                                                                          const double *__restrict__ rows1, int64_t ld, int n,
                                                                          double *__restrict__ gbest,
                                                                          const sx_state *__restrict__ state,
                                                                          double *__restrict__ share) {
    __shared__ double sf[kFinalThreads / kWave];
    if (state->done) return;
    const int64_t it = state->it + 1;  // the generation being finalised
    const double *__restrict__ src = ((it & 1) ? rows1 : rows0) + state->gbidx * ld;
    const int per = (((n + (int)gridDim.x - 1) / (int)gridDim.x + kFinalThreads - 1) / kFinalThreads) * kFinalThreads;
    const int e0 = (int)blockIdx.x * per, e1 = e0 + per < n ? e0 + per : n;
    double acc = 0.0;
    for (int eb = e0 + (int)threadIdx.x; eb < e1; eb += 8 * kFinalThreads) {
        double g[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * kFinalThreads;
            g[u] = e < e1 ? gbest[e] : 0.0;
            x[u] = e < e1 ? src[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * kFinalThreads;
            if (e < e1) {
                const double d = g[u] - x[u];
                acc += d * d;
                gbest[e] = x[u];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) sf[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ss = 0.0;
        for (int k = 0; k < kFinalThreads / kWave; ++k) ss += sf[k];
        share[blockIdx.x] = ss;
    }
}
__global__ __launch_bounds__(kWave) void wide_finalize_state_kernel(const double *__restrict__ share, int nshare,
                                                                    sx_state *__restrict__ state, int maxiter, double xtol,
                                                                    double ftol) {
    if (threadIdx.x != 0 || state->done) return;
    double ss = 0.0;
    for (int k = 0; k < nshare; ++k) ss += share[k];
    const double dx = sqrt(ss), bf = state->gfit;
    const int64_t it = state->it + 1;
    int status = SX_STATUS_NONE;
    if (dx <= xtol && bf <= ftol)
        status = 0;
    else if (bf <= ftol)
        status = 1;
    else if (it >= maxiter)
        status = -1;
    state->it = it;
    state->dx = dx;
    state->status = status;
    state->done = status != SX_STATUS_NONE;
}
static inline int wide_final_blocks(int64_t npart) { return npart < kWideFinalBlocks ? (int)npart : kWideFinalBlocks; }

extern "C" int sx_select_finalize(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                                  const double *rows1, int64_t ld, int n, double *gbest, sx_state *state, int maxiter,
                                  double xtol, double ftol, void *stream) {
    SX_REQUIRE(part_f && part_i && rows0 && rows1 && gbest && state && npart >= 1 && n >= 1,
               "sx_select_finalize: bad arguments");
    if (n > kMaxDim) {
        hipStream_t s = (hipStream_t)stream;
        const int nb = wide_final_blocks(npart);
        double *share = const_cast<double *>(part_f);  // (the records are consumed by the first launch: the header says so)
        hipLaunchKernelGGL(wide_finalize_best_kernel, dim3(1), dim3(kFinalThreads), 0, s, part_f, part_i, npart, state);
        hipLaunchKernelGGL(wide_finalize_row_kernel, dim3(nb), dim3(kFinalThreads), 0, s, rows0, rows1, ld, n, gbest, state, share);
        hipLaunchKernelGGL(wide_finalize_state_kernel, dim3(1), dim3(kWave), 0, s, share, nb, state, maxiter, xtol, ftol);
        SX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(select_finalize_kernel, dim3(1), dim3(kFinalThreads), 0, (hipStream_t)stream, part_f, part_i,
                       npart, rows0, rows1, ld, n, gbest, state, maxiter, xtol, ftol);
    SX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// Multi-GPU: shard best -> record, and best-of-generation over the gathered records.
// record = [ f, (double) global row, row[0..n) ]   (n + 2 doubles)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kFinalThreads) void shard_best_kernel(
    const double *__restrict__ part_f, const int64_t *__restrict__ part_i, int64_t npart,
    const double *__restrict__ rows0, const double *__restrict__ rows1, int64_t ld, int n,
    const sx_state *__restrict__ state, int64_t row0, double *__restrict__ record) {
    __shared__ double sf[kFinalThreads / kWave];
    __shared__ int64_t si[kFinalThreads / kWave];
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    scan_records(part_f, part_i, npart, bf, bi);
    const int64_t it = state->it + 1;  // the generation being finalised
    block_argmin(bf, bi, sf, si);
    const double *src = ((it & 1) ? rows1 : rows0) + bi * ld;
    final_copy(record + 2, src, n);
    if (threadIdx.x == 0) {
        record[0] = bf;
        record[1] = (double)(row0 + bi);  // exact below 2^53
    }
}

__global__ __launch_bounds__(kFinalThreads) void gather_finalize_kernel(const double *__restrict__ records, int world,
                                                                        int n, double *__restrict__ gbest,
                                                                        sx_state *__restrict__ state, int maxiter,
                                                                        double xtol, double ftol) {
    __shared__ double sf[kFinalThreads / kWave];
    if (state->done) return;
    const int64_t stride = n + 2;
    // every thread scans the (few) records: first minimum by (f, global row) = np.argmin over the whole population
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    int best = 0;
    for (int w = 0; w < world; ++w) {
        const double f = records[w * stride];
        const int64_t gi = (int64_t)records[w * stride + 1];
        if (f < bf || (f == bf && gi < bi)) {
            bf = f;
            bi = gi;
            best = w;
        }
    }
    const double *src = records + best * stride + 2;
    double acc = final_dist2(gbest, src, n);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) sf[threadIdx.x >> 6] = acc;
    __syncthreads();
    double ss = 0.0;
    for (int k = 0; k < kFinalThreads / kWave; ++k) ss += sf[k];
    const double dx = sqrt(ss);
    final_copy(gbest, src, n);
    if (threadIdx.x == 0) {
        const int64_t it = state->it + 1;
        int status = SX_STATUS_NONE;
        if (dx <= xtol && bf <= ftol)
            status = 0;
        else if (bf <= ftol)
            status = 1;
        else if (it >= maxiter)
            status = -1;
        state->it = it;
        state->gbidx = bi;
        state->gfit = bf;
        state->dx = dx;
        state->status = status;
        state->done = status != SX_STATUS_NONE;
    }
}

extern "C" int sx_shard_best(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                             const double *rows1, int64_t ld, int n, const sx_state *state, int64_t row0,
                             double *record, void *stream) {
    SX_REQUIRE(part_f && part_i && rows0 && rows1 && state && record && npart >= 1 && n >= 1,
               "sx_shard_best: bad arguments");
    hipLaunchKernelGGL(shard_best_kernel, dim3(1), dim3(kFinalThreads), 0, (hipStream_t)stream, part_f, part_i, npart,
                       rows0, rows1, ld, n, state, row0, record);
    SX_LAUNCH_CHECK();
    return 0;
}

extern "C" int sx_gather_finalize(const double *records, int world, int n, double *gbest, sx_state *state,
                                  int maxiter, double xtol, double ftol, void *stream) {
    SX_REQUIRE(records && gbest && state && world >= 1 && n >= 1, "sx_gather_finalize: bad arguments");
    hipLaunchKernelGGL(gather_finalize_kernel, dim3(1), dim3(kFinalThreads), 0, (hipStream_t)stream, records, world, n,
                       gbest, state, maxiter, xtol, ftol);
    SX_LAUNCH_CHECK();
    return 0;
}

namespace sx {
int add_finalize_node(hipGraph_t graph, hipGraphNode_t *prev, const double *part_f, const int64_t *part_i,
                      int64_t npart, const double *rows0, const double *rows1, int64_t ld, int n, double *gbest,
                      sx_state *state, int maxiter, double xtol, double ftol) {
    if (n > kMaxDim) {  // the three launches of sx_select_finalize's wide form
        int nb = wide_final_blocks(npart);
        double *share = const_cast<double *>(part_f);
        auto add = [&](void *fn, unsigned grid, unsigned block, void **ka) -> int {
            hipKernelNodeParams kq = {};
            kq.func = fn;
            kq.gridDim = dim3(grid);
            kq.blockDim = dim3(block);
            kq.sharedMemBytes = 0;
            kq.kernelParams = ka;
            kq.extra = nullptr;
            hipGraphNode_t nd;
            SX_HIP(hipGraphAddKernelNode(&nd, graph, *prev ? prev : nullptr, *prev ? 1 : 0, &kq));
            *prev = nd;
            return 0;
        };
        void *k1[] = {&part_f, &part_i, &npart, &state};
        if (int rc = add((void *)wide_finalize_best_kernel, 1, kFinalThreads, k1)) return rc;
        void *k2[] = {&rows0, &rows1, &ld, &n, &gbest, &state, &share};
        if (int rc = add((void *)wide_finalize_row_kernel, (unsigned)nb, kFinalThreads, k2)) return rc;
        void *k3[] = {&share, &nb, &state, &maxiter, &xtol, &ftol};
        return add((void *)wide_finalize_state_kernel, 1, kWave, k3);
    }
    void *kargs[] = {&part_f, &part_i, &npart, &rows0, &rows1, &ld, &n, &gbest, &state, &maxiter, &xtol, &ftol};
    hipKernelNodeParams kp = {};
    kp.func = (void *)select_finalize_kernel;
    kp.gridDim = dim3(1);
    kp.blockDim = dim3(kFinalThreads);
    kp.sharedMemBytes = 0;
    kp.kernelParams = kargs;
    kp.extra = nullptr;
    hipGraphNode_t node;
    SX_HIP(hipGraphAddKernelNode(&node, graph, *prev ? prev : nullptr, *prev ? 1 : 0, &kp));
    *prev = node;
    return 0;
}
}  // namespace sx

// ---------------------------------------------------------------------------
// Initial population in Philox mode: the reference's Latin hypercube (_common.py:109-120)
//     u = rand(P, n) / P + linspace(-1, 1, P, endpoint=False)[:, None];  column j permuted by permutation(P);
//     pop = u * 0.5 (upper - lower) + 0.5 (upper + lower)
// with counter-based draws, so that every row can be produced anywhere (a rank draws only ITS rows; nothing is
// built on the host): row i of column j takes stratum sigma_j(i), sigma_j a keyed bijection of [0, P) --
// three rounds of x -> (x * m + a) mod 2^b, x ^= x >> ceil(b/2) on b = bit_length(P - 1) bits (m odd; keys from
// Philox calls (slot j, purpose kPurposeInitPerm)), cycle-walked into [0, P) -- and a 53-bit uniform keyed by
// (global row, element) inside the stratum.  Same strata, same half-cell jitter, same scaling arithmetic
// (two roundings each) as the reference; oracle counterpart: oracle/streams.py PhiloxStream.lhs_population.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void philox_lhs_kernel(double *__restrict__ X, int64_t rows, int n, int64_t ld,
                                                          int64_t row0, int64_t P, const double *__restrict__ lower,
                                                          const double *__restrict__ upper, uint32_t k0, uint32_t k1) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * n) return;
    const int64_t r = t / n;
    const int e = (int)(t % n);
    const uint64_t gi = (uint64_t)(row0 + r);
    // the column's permutation keys
    const U4 ka = philox4x32_10((uint32_t)e, 0u, 0u, kPurposeInitPerm, k0, k1);
    const U4 kb = philox4x32_10((uint32_t)e, 1u, 0u, kPurposeInitPerm, k0, k1);
    const uint64_t m[3] = {(uint64_t)(ka.x | 1u), (uint64_t)(ka.y | 1u), (uint64_t)(ka.z | 1u)};
    const uint64_t ad[3] = {(uint64_t)kb.x, (uint64_t)kb.y, (uint64_t)kb.z};
    int b = 1;
    while (((uint64_t)1 << b) < (uint64_t)P) ++b;  // P >= 2: b = bit_length(P - 1)
    const uint64_t mask = ((uint64_t)1 << b) - 1u;
    const int sh = (b + 1) / 2;
    uint64_t x = gi;
    do {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x = (x * m[k] + ad[k]) & mask;
            x ^= x >> sh;
        }
    } while (x >= (uint64_t)P);
    const double u = philox_u53(e, lanes_per_row(n), (uint32_t)gi, 0u, kPurposeInitJitter, k0, k1);
    const double step = 2.0 / (double)P;                      // np.linspace(-1, 1, P, endpoint=False): arange * step + start
    const double v = u / (double)P + ((double)x * step + -1.0);
    const double lo = lower[e], hi = upper[e];
    X[r * ld + e] = v * (0.5 * (hi - lo)) + 0.5 * (hi + lo);
}

extern "C" int sx_philox_lhs(double *X, int64_t rows, int n, int64_t ld, int64_t row0, int64_t P, const double *lower,
                             const double *upper, uint32_t key0, uint32_t key1, void *stream) {
    SX_REQUIRE(X && lower && upper, "sx_philox_lhs: null pointer");
    SX_REQUIRE(rows >= 1 && n >= 1 && ld >= n && row0 >= 0 && P >= 2 && row0 + rows <= P && P < ((int64_t)1 << 31),
               "sx_philox_lhs: bad shape");
    const int64_t tot = rows * n;
    hipLaunchKernelGGL(philox_lhs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, rows,
                       n, ld, row0, P, lower, upper, key0, key1);
    SX_LAUNCH_CHECK();
    return 0;
}
