// Wide rows: individuals of more than kWideFrom = 2048 elements (sx_device.hpp; the wavefront-per-row kernels can serve up to
// kMaxDim = 4096 but lose to these from there on: profiles/r5_wide_threshold2.txt), ONE WORKGROUP per individual.
//
// The reference has no dimension limit (stochopy/optimize/de/_de.py:208-218: rows are (n,) numpy vectors of any length;
// stochopy/optimize/vdcma/_vdcma.py:144-458 exists for n >= 4096); the row kernels of sx_rowops.hpp do: one wavefront per
// row, the whole trial vector in that wavefront's LDS slice, numpy's summation plan in the kernel arguments (<= 64
// leaves).  Here
//   * numpy does not hand a long reduction to its pairwise sum in one piece: the ufunc machinery feeds the inner loop
//     np.getbufsize() = 8192 elements at a time, so `a.sum()` of m > 8192 terms is
//         acc = 0.0;  for every 8192-term piece, in order:  acc = acc + pairwise_sum(piece)
//     (checked against numpy for m = 8193 ... 65536, and against the reference's own objective values:
//     tests/golden/vdcma_wide.json objective_kat).  Rows of up to 4096 elements never see this; wide rows do;
//   * the plan lives in device memory (built once per number of terms, cached): for every 8192-term piece the leaf table
//     of numpy's pairwise recursion (loops_utils.h.src: <= 128 terms per leaf, split at n/2 rounded down to a multiple of
//     8) and the recursion's combines ordered by LEVEL -- combines of one level touch disjoint leaf slots, so a level is
//     one parallel step (6 steps for a full piece instead of 63 dependent additions); the pieces' sums (a full piece has
//     64 leaves: piece c ends in slot 64 c) are then added up in order;
//   * the row is walked in CHUNKS of whole leaves (as many as fit 4096 elements: the plan's chunk table): the workgroup's
//     threads produce the chunk's elements (coalesced runs, four elements and all their loads in flight per thread, the Philox
//     layout of the whole-wave rows: element e is lane e % 64 at step e / 64), stage them in LDS, and every 8-lane group takes
//     one leaf (eight accumulators over the 8-blocks, the tree, the tail) exactly as row_reduce_leaves_fused does: same
//     additions in the same order, same bits;
//   * the recursion's combines: one wavefront per 8192-term piece, leaf slot l on lane l, the levels walked with lane shuffles
//     (wide_finish), then add.reduce's own loop over the pieces;
//   * a row whose stage fits twice on a CU (72 KB: up to ~8 900 elements) stays RESIDENT in LDS (one chunk: the DE trial is
//     stored from there when it wins, PSO's new position is copied to pbest from there); longer rows are STREAMED through the
//     4096-element stage: DE writes the trial into the next buffer as it is produced and copies the old row over it if the
//     trial lost, PSO re-reads the position it has just written.
// What bounds these kernels is vector-instruction issue and the number of workgroups a CU holds, not the LDS or the bytes in
// flight (counters: profiles/r5_wide_ab3.txt ... r5_wide_ab5.txt) -- hence: strategy-specialised DE kernels (63 VGPRs for
// Rosenbrock best1bin: four workgroups per CU), unpredicated code for full chunks and full leaves, a select instead of a
// branch per term in short leaves (cheap terms only), tail terms by lane.
// Objective values, draws, selection and records are those of the narrow kernels: a wide run is compared with the oracle
// bit for bit (tests/test_gpu_wide.py).
//
// Reference code replaced: as sx_de.hip / sx_pso.hip / sx_core.hip (de/_de.py:314-351, cpso/_cpso.py:324-361,
// _common.py:34-90, 123-130; factory/benchmark.py:14-156).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "sx_device.hpp"
#include "sx_host.hpp"
#include "sx_rowops.hpp"
#include "sx_wide.hpp"

using namespace sx;

namespace {

// ---------------------------------------------------------------------------
// the plan in device memory:  [0] nleaf  [1] tail  [2] mb  [3] nlevels  [4] npiece (8192-term pieces of the sum)
//                             [5] where the last piece's partner words start   [6] where the chunk table starts
//                             [7 .. 7+nleaf)            end block (exclusive) of leaf t
//                             [.. + nlevels + 1)        first combine of level v (prefix sums)
//                             [.. + 2 (nleaf-npiece))   (left, right) leaf slots of the combines, level by level
//                             [.. + 128)                the LAST piece's combines as seen by its leaves: word w of leaf slot
//                                                       l (slot within the piece) at [64 w + l], byte v & 3 of word v >> 2 =
//                                                       the slot (<= 64) whose sum leaf l takes in at level v, 0xff: none
//                             [.. + 2 + nchunk)         nchunk, then the first leaf of every chunk of a STREAMED row and nleaf:
//                                                       as many whole leaves as fit kChunkElems elements.  (A fixed 32 leaves
//                                                       per chunk left Rosenbrock rows of 2^k elements -- 2^k - 1 terms: 33
//                                                       or 65 leaves in the last piece -- with a chunk of ONE leaf.)
// ---------------------------------------------------------------------------
struct HostPlan {
    std::vector<int32_t> end;
    std::vector<std::pair<int32_t, int32_t>> merge;
    std::vector<int32_t> level;
    int blocks = 0;
    // returns (slot of the sum, height of the subtree)
    std::pair<int, int> rec(int64_t m) {
        if (m <= 128) {
            blocks += (int)(m / 8);
            end.push_back(blocks);
            return {(int)end.size() - 1, 0};
        }
        int64_t h = m / 2;
        h -= h % 8;
        const auto a = rec(h);
        const auto b = rec(m - h);
        const int lv = a.second > b.second ? a.second : b.second;
        merge.push_back({a.first, b.first});
        level.push_back(lv);
        return {a.first, lv + 1};
    }
};

constexpr int64_t kNumpyBuf = 8192;  // np.getbufsize(): the pieces numpy's reduction hands to its pairwise sum
constexpr int kPieceLeaves = 64;     // leaves of a full piece (128 terms each)
constexpr int kPlanHeader = 7;
constexpr int kChunkElems = 4096;    // elements of a streamed row staged at a time (kStageElems below has the slack)

struct CachedPlan {
    int32_t *dev = nullptr;
    int nleaf = 0;
};
std::mutex g_plan_mutex;
std::map<std::pair<int, int64_t>, CachedPlan> g_plans;  // (device, terms) -> plan; never freed (a few KB each)

int get_plan(int64_t m, hipStream_t s, CachedPlan *out) {
    int dev = 0;
    SX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    auto it = g_plans.find({dev, m});
    if (it != g_plans.end()) {
        *out = it->second;
        return 0;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        set_error("wide rows: the summation plan of this row length is built (allocation + upload) on first use, which "
                  "cannot happen inside a stream capture -- evaluate once before capturing");
        return -1;
    }
    HostPlan hp;
    int npiece = 0;
    for (int64_t c0 = 0; c0 < m; c0 += kNumpyBuf, ++npiece) {  // (a piece starts at a block boundary and at leaf slot 64 c)
        hp.blocks = (int)(c0 / kGroup);
        (void)hp.rec(m - c0 < kNumpyBuf ? m - c0 : kNumpyBuf);
    }
    const int nleaf = (int)hp.end.size();
    int nlevels = 0;
    for (int lv : hp.level) nlevels = lv + 1 > nlevels ? lv + 1 : nlevels;
    const size_t partner_at = kPlanHeader + (size_t)nleaf + (size_t)nlevels + 1 + 2 * hp.merge.size();
    std::vector<int32_t> chunks;  // first leaves of the chunks
    for (int t0 = 0; t0 < nleaf;) {
        const int64_t e0 = t0 > 0 ? (int64_t)hp.end[t0 - 1] * kGroup : 0;
        int t = t0 + 1;  // (a leaf has at most 128 terms + the row's tail of 7)
        while (t < nleaf && (t + 1 == nleaf ? m : (int64_t)hp.end[t] * kGroup) - e0 <= kChunkElems) ++t;
        chunks.push_back(t0);
        t0 = t;
    }
    chunks.push_back(nleaf);
    const size_t chunk_at = partner_at + 2 * kPieceLeaves;
    std::vector<int32_t> buf(chunk_at + 1 + chunks.size());
    buf[6] = (int32_t)chunk_at;
    buf[chunk_at] = (int32_t)chunks.size() - 1;
    for (size_t k = 0; k < chunks.size(); ++k) buf[chunk_at + 1 + k] = chunks[k];
    buf[0] = nleaf;
    buf[1] = (int32_t)(m % 8);
    buf[2] = (int32_t)(m / 8);
    buf[3] = nlevels;
    buf[4] = npiece;
    buf[5] = (int32_t)partner_at;
    for (int t = 0; t < nleaf; ++t) buf[kPlanHeader + t] = hp.end[t];
    int32_t *off = buf.data() + kPlanHeader + nleaf;
    int32_t *pairs = off + nlevels + 1;
    int pos = 0;
    for (int lv = 0; lv < nlevels; ++lv) {
        off[lv] = pos;
        for (size_t k = 0; k < hp.merge.size(); ++k) {
            if (hp.level[k] != lv) continue;
            pairs[2 * pos] = hp.merge[k].first;
            pairs[2 * pos + 1] = hp.merge[k].second;
            ++pos;
        }
    }
    off[nlevels] = pos;
    {  // the last piece's combines, per leaf slot and level.  A piece of m <= 8192 terms has at most 65 leaves and 7 levels
       // (enumerated: 65 / 7 for 441 values of m, e.g. 8191 -- a node of 129..135 terms splits once more); a 65th leaf is
       // the piece's last one and only ever the RIGHT operand of one combine (r = 64).
        uint32_t *pw = (uint32_t *)buf.data() + partner_at;
        for (int i = 0; i < 2 * kPieceLeaves; ++i) pw[i] = 0xffffffffu;
        const int first = (npiece - 1) * kPieceLeaves;
        for (size_t k = 0; k < hp.merge.size(); ++k) {
            const int l = hp.merge[k].first - first, r = hp.merge[k].second - first, lv = hp.level[k];
            if (l < 0) continue;  // a combine of an earlier (full) piece
            if (lv >= 7 || l >= kPieceLeaves || r <= l || r > kPieceLeaves) {
                set_error("wide rows: the summation plan of this row length does not fit the per-piece form");
                return -1;
            }
            uint32_t &w = pw[(lv >> 2) * kPieceLeaves + l];
            w = (w & ~(0xffu << (8 * (lv & 3)))) | ((uint32_t)r << (8 * (lv & 3)));
        }
    }
    CachedPlan cp;
    cp.nleaf = nleaf;
    SX_HIP(hipMalloc((void **)&cp.dev, buf.size() * sizeof(int32_t)));
    SX_HIP(hipMemcpy(cp.dev, buf.data(), buf.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    g_plans[{dev, m}] = cp;
    *out = cp;
    return 0;
}

struct WideCtx {
    const int32_t *end, *lvl, *pairs, *chunk;  // chunk[0]: chunks of a streamed row, chunk[1 + k]: first leaf of chunk k
    int nleaf, tail, mb, nlevels, npiece;
    uint32_t pw0, pw1;  // this lane's partner words of the last piece (loaded here, needed at the very end)
};
__device__ __forceinline__ WideCtx wide_ctx(const int32_t *__restrict__ plan) {
    WideCtx c;
    c.nleaf = plan[0], c.tail = plan[1], c.mb = plan[2], c.nlevels = plan[3], c.npiece = plan[4];
    const uint32_t *pw = (const uint32_t *)plan + plan[5] + ((int)threadIdx.x & (kWave - 1));
    c.pw0 = pw[0], c.pw1 = pw[kPieceLeaves];
    c.chunk = plan + plan[6];
    c.end = plan + kPlanHeader;
    c.lvl = c.end + c.nleaf;
    c.pairs = c.lvl + c.nlevels + 1;
    return c;
}

// Where element e of the row sits in the stage.  A leaf of numpy's recursion is usually 128 elements = 1 KB, and the 8-lane
// groups of a wavefront read consecutive leaves at the same block: four groups of a ds_read_b64 half on the same 16 banks
// (a 4-way conflict on every read).  Eight doubles of padding per 128 elements move consecutive leaves 64 B apart in the
// banks.  So that "the next element" stays ONE slot away for the reduction (Rosenbrock-like terms), the first element of
// every 128-segment is staged twice: in its place and in the pad slot right after its predecessor.
// Measured (profiles/r5_wide_ab3.txt, sx_eval Rosenbrock n = 4096): SQ_LDS_BANK_CONFLICT 23.8 M -> 1.2 M cycles per launch, and the
// launch 4 % SLOWER (the address arithmetic: 96 M -> 117 M vector instructions) -- these kernels are bound by vector-instruction
// issue, not by the LDS.  Kept behind the switch, off.
#ifndef SX_WIDE_PAD
#define SX_WIDE_PAD 0
#endif
__host__ __device__ __forceinline__ int stage_pos(int e) { return SX_WIDE_PAD ? e + ((e >> 7) << 3) : e; }
template <bool NEXT>
__device__ __forceinline__ void stage_put(double *Sd, int e, int e0, double v) {  // e0: the first element of the chunk
    const int p = stage_pos(e);
    Sd[p] = v;
    if (SX_WIDE_PAD && NEXT && (e & 127) == 0 && e > e0) Sd[p - 8] = v;
}
#ifndef SX_WIDE_LEAF_BATCH
#define SX_WIDE_LEAF_BATCH(FUN) 8  // (4 for the cosine objectives: 100 -> 95 VGPRs, one variant spills: not taken)
#endif
// Leaves [leaf0, leaf1) of the row, elements staged at Sd[e] (Sd = stage - first staged element): one leaf per 8-lane
// group and pass -- lane j walks accumulator j over the leaf's 8-blocks, forming the terms on the way
// (row_reduce_leaves_fused's arithmetic) -> LA[t] (and LB[t]).
template <int FUN, int T>
__device__ __forceinline__ void wide_reduce_leaves(const double *Sd, int leaf0, int leaf1, const WideCtx &c, double *LA,
                                                   double *LB) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    const int j = (int)threadIdx.x & (kGroup - 1), grp = (int)threadIdx.x >> 3;
    const double identB = BMUL ? 1.0 : 0.0;
    for (int t0 = leaf0; t0 < leaf1; t0 += T / kGroup) {
        // (a pass that is one leaf over -- 33 or 65 leaves: Rosenbrock rows of 2^k elements -- costs one wavefront, not all)
        if (t0 + (grp & ~(kWave / kGroup - 1)) >= leaf1) break;
        const bool on = t0 + grp < leaf1;
        const int t = on ? t0 + grp : leaf1 - 1;  // idle groups shadow the last leaf and keep nothing
        const int b0 = t > 0 ? c.end[t - 1] : 0, b1 = c.end[t];
        const int cnt = b1 - b0;  // 8 .. 16 blocks (fewer only in a last piece of < 64 terms; none: its terms are all tail)
        const double *UA = Sd + stage_pos(b0 * kGroup) + j, *UB = UA + 8;  // UB: blocks past the leaf's 128-boundary
        const int ustar = 16 - (b0 & 15);
        double chA = 0.0, chB = identB;
        constexpr int B = SX_WIDE_LEAF_BATCH(FUN);  // blocks whose elements are in flight together
        if (light_objective<FUN>() && __all(cnt == kLeafBlocks)) {
            // every leaf of this wavefront has its 16 blocks (the usual case: 128-term leaves): no predicates.  (Not for the
            // cosine objectives: their terms dwarf the predicates, and eight unpredicated cosines side by side take 176-251
            // VGPRs.)
#pragma unroll
            for (int h0 = 0; h0 < kLeafBlocks; h0 += B) {
                double x[B], xn[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const double *U = (SX_WIDE_PAD && h0 + u >= ustar) ? UB : UA;
                    x[u] = U[(h0 + u) * kGroup];
                    xn[u] = O::NEXT ? U[(h0 + u) * kGroup + 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    double a, b;
                    O::term(x[u], xn[u], (b0 + h0 + u) * kGroup + j, a, b);
                    if (h0 + u == 0) {
                        chA = a;
                        chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        } else {
            // some leaf of this wavefront is shorter: the blocks past a leaf's end are read all the same (they lie inside
            // the workgroup's LDS: the stage's slack or the leaf sums behind it) and their terms are dropped from the chain
#pragma unroll
            for (int h0 = 0; h0 < kLeafBlocks; h0 += B) {
                double x[B], xn[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const double *U = (SX_WIDE_PAD && h0 + u >= ustar) ? UB : UA;
                    x[u] = U[(h0 + u) * kGroup];
                    xn[u] = O::NEXT ? U[(h0 + u) * kGroup + 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const bool in = h0 + u < cnt;
                    double a, b;
                    if constexpr (light_objective<FUN>()) {
                        O::term(x[u], xn[u], (b0 + h0 + u) * kGroup + j, a, b);
                        if (h0 + u == 0) {
                            chA = in ? a : 0.0;
                            chB = in ? b : identB;
                        } else {
                            chA = in ? chA + a : chA;
                            if (TWO) chB = in ? combine<BMUL>(chB, b) : chB;
                        }
                    } else if (in) {  // the cosine objectives: one term after the other (side by side they take 150-230 VGPRs)
                        O::term(x[u], xn[u], (b0 + h0 + u) * kGroup + j, a, b);
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
        }
        double curA = group_tree<false>(chA);
        double curB = TWO ? group_tree<BMUL>(chB) : identB;
        if (t == c.nleaf - 1 && c.tail > 0) {
            // the tail terms (fewer than 8), added one by one after the last leaf's tree: lane k of the group forms term k,
            // the sums take them in order (one term's arithmetic instead of `tail` times that, for the whole wavefront)
            const int e0 = c.mb * kGroup;
            double ta = 0.0, tb = identB;
            if (j < c.tail) {
                const double *te = Sd + stage_pos(e0 + j);
                O::term(te[0], O::NEXT ? te[1] : 0.0, e0 + j, ta, tb);
            }
            const int g0 = (int)threadIdx.x & (kWave - kGroup);  // first lane of this group within the wavefront
#pragma unroll
            for (int k = 0; k < kGroup - 1; ++k) {
                const double va = __shfl(ta, g0 + k, kWave);
                if (k < c.tail) curA = curA + va;
                if (TWO) {
                    const double vb = __shfl(tb, g0 + k, kWave);
                    if (k < c.tail) curB = combine<BMUL>(curB, vb);
                }
            }
        }
        if (on && j == 0) {
            LA[t] = curA;
            if (TWO) LB[t] = curB;
        }
    }
}

// The recursion's combines.  A piece (8192 terms) has 64 leaves (a last, shorter one up to 65): ONE WAVEFRONT takes a piece, leaf slot l on lane l,
// and walks the levels of the recursion with lane shuffles -- at level v lane l takes in the sum of the slot the plan
// names (a full piece: the perfect tree, l + 2^v into l when l is a multiple of 2^(v+1); the last piece: the partner
// words) -- the same additions in the same tree as numpy's recursion, without a barrier or a table look-up per level
// (the level-by-level form over all pieces in LDS -- SX_WIDE_FINISH_SHFL=0 -- costs a workgroup of a 4096-element row
// about as long as loading the row).  Then add.reduce's own loop: the identity, then the pieces in order.
#ifndef SX_WIDE_FINISH_SHFL
#define SX_WIDE_FINISH_SHFL 1
#endif
template <int FUN, int T>
__device__ __forceinline__ double wide_finish(const WideCtx &c, int n, double *LA, double *LB) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    __syncthreads();
#if SX_WIDE_FINISH_SHFL
    const int lane = (int)threadIdx.x & (kWave - 1);
    for (int p = (int)threadIdx.x >> 6; p < c.npiece; p += T / kWave) {
        const bool lastp = p == c.npiece - 1;
        const int nl = lastp ? c.nleaf - p * kPieceLeaves : kPieceLeaves;
        double a = lane < nl ? LA[p * kPieceLeaves + lane] : 0.0;
        double b = (TWO && lane < nl) ? LB[p * kPieceLeaves + lane] : (BMUL ? 1.0 : 0.0);
#pragma unroll
        for (int lv = 0; lv < 7; ++lv) {
            const int named = (int)(((lv < 4 ? c.pw0 : c.pw1) >> (8 * (lv & 3))) & 0xffu);
            const int full = (lv < 6 && (lane & ((2 << lv) - 1)) == 0) ? lane + (1 << lv) : 0xff;
            const int r = lastp ? named : full;
            const bool on = r != 0xff, extra = r == kPieceLeaves;  // (a last piece's 65th leaf: read where it lies)
            double ra = __shfl(a, on ? (r & (kWave - 1)) : lane, kWave);
            if (extra) ra = LA[p * kPieceLeaves + kPieceLeaves];
            if (on) a = a + ra;
            if (TWO) {
                double rb = __shfl(b, on ? (r & (kWave - 1)) : lane, kWave);
                if (extra) rb = LB[p * kPieceLeaves + kPieceLeaves];
                if (on) b = combine<BMUL>(b, rb);
            }
        }
        if (lane == 0) {
            LA[p * kPieceLeaves] = a;
            if (TWO) LB[p * kPieceLeaves] = b;
        }
    }
    __syncthreads();
#else
    for (int lv = 0; lv < c.nlevels; ++lv) {
        const int o1 = c.lvl[lv + 1];
        for (int i = c.lvl[lv] + (int)threadIdx.x; i < o1; i += T) {
            const int l = c.pairs[2 * i], r = c.pairs[2 * i + 1];
            LA[l] = LA[l] + LA[r];
            if (TWO) LB[l] = combine<BMUL>(LB[l], LB[r]);
        }
        __syncthreads();
    }
#endif
    // add.reduce starts from the identity and takes the 8192-term pieces in order (piece c: slot 64 c)
    double sa = 0.0, sb = BMUL ? 1.0 : 0.0;
    for (int p = 0; p < c.npiece; ++p) {
        sa = sa + LA[p * kPieceLeaves];
        if (TWO) sb = combine<BMUL>(sb, LB[p * kPieceLeaves]);
    }
    return O::finish(sa, sb, n);
}

// The row, chunk by chunk: produce(e0, e1, e1s, Sd) stages the elements [e0, e1s) with stage_put(Sd, e, e0, .) and commits whatever it
// writes to memory for [e0, e1) only -- for an objective that reads the next element too, e1s = e1 + 1: that element is
// staged again (and committed) by the next chunk.  chunk_leaves > 0: the row is resident (one chunk); 0: streamed through the plan's chunks.
__device__ __forceinline__ void wide_chunk(const WideCtx &c, int n, bool next, bool resident, int k, int &leaf0, int &leaf1,
                                           int &e0, int &e1, int &e1s) {
    leaf0 = resident ? 0 : c.chunk[1 + k];
    leaf1 = resident ? c.nleaf : c.chunk[2 + k];
    const bool last = leaf1 == c.nleaf;
    e0 = leaf0 > 0 ? c.end[leaf0 - 1] * kGroup : 0;
    e1 = last ? n : c.end[leaf1 - 1] * kGroup;
    e1s = last ? n : e1 + (next ? 1 : 0);
}
template <int FUN, int T, class Produce>
__device__ __forceinline__ double wide_row(const WideCtx &c, int n, int chunk_leaves, double *S, double *LA, double *LB,
                                           Produce &&produce) {
    constexpr bool NEXT = Obj<FUN>::NEXT;
    const bool resident = chunk_leaves > 0;
    const int nchunk = resident ? 1 : c.chunk[0];
    for (int k = 0; k < nchunk; ++k) {
        int leaf0, leaf1, e0, e1, e1s;
        wide_chunk(c, n, NEXT, resident, k, leaf0, leaf1, e0, e1, e1s);
        if (k > 0) __syncthreads();  // the previous chunk's leaves have been read
        produce(e0, e1, e1s, S - stage_pos(e0));
        __syncthreads();
        wide_reduce_leaves<FUN, T>(S - stage_pos(e0), leaf0, leaf1, c, LA, LB);
    }
    return wide_finish<FUN, T>(c, n, LA, LB);
}

constexpr int kEvalThreads = 256;
#ifndef SX_WIDE_EVAL_PIPE_FROM
#define SX_WIDE_EVAL_PIPE_FROM 8192  // rows longer than this (several chunks) of the light objectives take the pipelined form
#endif
constexpr int kGenThreads = 512;
// A resident row + its leaf sums must fit here: TWO workgroups per CU.  (Up to 148 KB -- one workgroup per CU, rows of up to
// ~18 000 elements -- was the first form; with one workgroup on the CU the phases of a row run one after the other: n = 12 000 ...
// 18 000 streamed instead: DE 0.53-0.58 -> 0.60-0.64 of the HBM peak, PSO 0.48-0.52 -> 0.52-0.64, although a streamed DE row
// that loses is copied back; a prefetching loop and 1024-thread workgroups for the one-workgroup case reached 0.54-0.58 / 0.52.
// At n = 8192 -- two resident workgroups per CU -- resident wins: DE Rastrigin 0.65 against 0.55, PSO Ackley 0.62 against 0.57.
// profiles/r5_wide_resident_ab.txt, r5_wide_onewg.txt.)
constexpr size_t kResidentLds = 72 * 1024;
constexpr int kStageElems = kChunkElems + 16 + 128 + (SX_WIDE_PAD ? 8 * (kChunkElems / 128 + 2) : 0);

__host__ __device__ inline int wide_leaf_cap(int n) { return n / 64 + 2; }
// (+ 128: the blocks a short last leaf does not have are read all the same, wide_reduce_leaves)
__host__ __device__ inline int wide_resident_elems(int n) { return n + 16 + 128 + (SX_WIDE_PAD ? 8 * (n / 128 + 2) : 0); }
inline size_t wide_lds_bytes(int n, bool resident) {
    return ((size_t)(resident ? wide_resident_elems(n) : kStageElems) + 2 * (size_t)wide_leaf_cap(n)) * sizeof(double);
}
inline size_t wide_resident_limit() {  // SX_WIDE_RESIDENT_KB: measurement hook (tools/bench_wide.py)
    static const size_t lim = [] {
        const char *e = getenv("SX_WIDE_RESIDENT_KB");
        const size_t v = e != nullptr ? (size_t)atol(e) * 1024 : kResidentLds;
        return v < kResidentLds ? v : kResidentLds;
    }();
    return lim;
}
inline bool wide_resident(int n) { return wide_lds_bytes(n, true) <= wide_resident_limit(); }

// ---------------------------------------------------------------------------
// objective of rows of X (sx_eval; the CMA-ES family's un-standardisation / clipping / penalty sums as in eval_kernel)
// ---------------------------------------------------------------------------
template <int FUN, bool PIPE>
__global__ __launch_bounds__(kEvalThreads) void wide_eval_kernel(const double *__restrict__ X, int64_t P, int n, int64_t ldx,
                                                                 const double *__restrict__ xm, const double *__restrict__ xstd,
                                                                 double *__restrict__ f, const int32_t *__restrict__ plan,
                                                                 double *__restrict__ part_f, int64_t *__restrict__ part_i,
                                                                 const int clip, const double *__restrict__ pen_v,
                                                                 double *__restrict__ pen_out) {
    constexpr int T = kEvalThreads;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s_pen[T / kWave];
    const WideCtx c = wide_ctx(plan);
    double *S = lds, *LA = lds + kStageElems, *LB = LA + wide_leaf_cap(n);
    const int64_t row = blockIdx.x;
    const double *__restrict__ xr = X + row * ldx;
    const bool affine = xm != nullptr;
    const int tid = (int)threadIdx.x;
    double pacc = 0.0;
    // The chunks, one after the other: fetch (16 row elements per thread, all their loads in flight; + the look-ahead
    // element of an objective that reads its neighbour, thread 0's 17th) -> commit to the stage -> the chunk's leaves.
    // PIPE: the loads of chunk k + 1 are issued BEFORE the leaves of chunk k are reduced, so a workgroup always has 32 KB on
    // its way; costs 32-36 VGPRs: measured +4 ... +9 % for Rosenbrock / Sphere rows of 16 384 and 65 536 elements, -3 ... -10 %
    // for one- and two-chunk rows and for the cosine objectives (three workgroups per CU instead of four) --
    // profiles/r5_wide_ab2.txt; hence PIPE only for light objectives and n > SX_WIDE_EVAL_PIPE_FROM.  Either way the same
    // elements are staged at the same places and the same leaves are formed: same bits.
    // A chunk that has all its 16 x 256 elements (every chunk but a row's last) and a plain evaluation (no clipping, no
    // un-standardisation: sx_eval's own calls) take unpredicated code: the general form spends ~23 vector instructions per
    // ELEMENT on its run-time switches, as much as the objective itself (counters: profiles/r5_wide_ab3.txt).
    constexpr bool NEXT = Obj<FUN>::NEXT;
    constexpr int NV = kChunkElems / T;  // 16
    static_assert(NV * T == kChunkElems, "a chunk is a whole number of elements per thread");
    const bool plain = !clip && !affine;
    double xv[NV], xlook = 0.0;
    auto fetch = [&](int e0, int e1s) {
        if (e1s - e0 >= NV * T) {
#pragma unroll
            for (int u = 0; u < NV; ++u) xv[u] = xr[e0 + tid + u * T];
        } else {
#pragma unroll
            for (int u = 0; u < NV; ++u) xv[u] = e0 + tid + u * T < e1s ? xr[e0 + tid + u * T] : 0.0;
        }
        if (NEXT) xlook = (tid == 0 && e0 + NV * T < e1s) ? xr[e0 + NV * T] : 0.0;
    };
    auto put = [&](int e, int e0c, int e1, double v, double *Sd) {
        if (clip) {  // cmaes/_constraints.py:29-31, :79
            const double cl = v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
            if (pen_v != nullptr && e < e1) pacc += ((cl - v) * (cl - v)) * pen_v[e];
            v = cl;
        }
        if (affine) v = v * xstd[e] + xm[e];  // cmaes/_cmaes.py:171
        stage_put<NEXT>(Sd, e, e0c, v);
    };
    const int nchunk = c.chunk[0];
    int leaf0, leaf1, e0, e1, e1s;
    wide_chunk(c, n, NEXT, false, 0, leaf0, leaf1, e0, e1, e1s);
    if (PIPE) fetch(e0, e1s);
    for (int k = 0; k < nchunk; ++k) {
        wide_chunk(c, n, NEXT, false, k, leaf0, leaf1, e0, e1, e1s);
        double *Sd = S - stage_pos(e0);
        if (!PIPE) fetch(e0, e1s);
        if (plain && !SX_WIDE_PAD) {
            if (e1s - e0 >= NV * T) {
#pragma unroll
                for (int u = 0; u < NV; ++u) Sd[e0 + tid + u * T] = xv[u];
            } else {
#pragma unroll
                for (int u = 0; u < NV; ++u)
                    if (e0 + tid + u * T < e1s) Sd[e0 + tid + u * T] = xv[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < NV; ++u)
                if (e0 + tid + u * T < e1s) put(e0 + tid + u * T, e0, e1, xv[u], Sd);
        }
        if (NEXT && tid == 0 && e0 + NV * T < e1s) put(e0 + NV * T, e0, e1, xlook, Sd);
        for (int e = e0 + NV * T + (NEXT ? 1 : 0) + tid; e < e1s; e += T) put(e, e0, e1, xr[e], Sd);  // (a last chunk's tail terms)
        __syncthreads();
        const bool more = k + 1 < nchunk;
        if (PIPE && more) {
            int nl0, nl1, ne0, ne1, ne1s;
            wide_chunk(c, n, NEXT, false, k + 1, nl0, nl1, ne0, ne1, ne1s);
            fetch(ne0, ne1s);
        }
        wide_reduce_leaves<FUN, T>(Sd, leaf0, leaf1, c, LA, LB);
        if (more) __syncthreads();  // the chunk's leaves have been read
    }
    const double val = wide_finish<FUN, T>(c, n, LA, LB);
    if (pen_out != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pacc += __shfl_xor(pacc, off, kWave);
        if ((tid & 63) == 0) s_pen[tid >> 6] = pacc;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < T / kWave; ++w) s += s_pen[w];
            pen_out[row] = s;
        }
    }
    if (tid == 0) {
        f[row] = val;
        if (part_f != nullptr) part_f[row] = val, part_i[row] = row;
    }
}

// ---------------------------------------------------------------------------
// DE generation (de_generation_kernel<XM = 0> for wide rows): the state is ONE sx_state, the best / termination step a
// kernel of its own; one record per row.
// ---------------------------------------------------------------------------
// STRAT >= 0: best1bin / rand1bin with constraints=None at compile time (two or three donor rows in flight instead of five,
// no bounds / resample registers: the generic form's 136-148 VGPRs leave ONE 512-thread workgroup per CU).
template <int FUN, int RNG, int STRAT = -1>
__global__ __launch_bounds__(kGenThreads, 4) void wide_de_kernel(const sx_de_args a, const int32_t *__restrict__ plan,
                                                              const int chunk_leaves) {
    constexpr int T = kGenThreads;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const sx_state *sin = a.state;
    if (sin->done) return;
    const WideCtx c = wide_ctx(plan);
    const int n = a.n;
    const bool resident = chunk_leaves > 0;
    double *S = lds, *LA = lds + (resident ? wide_resident_elems(n) : kStageElems), *LB = LA + wide_leaf_cap(n);
    const int64_t P = a.P, ld = a.ld, row = blockIdx.x, it = sin->it;
    const int tid = (int)threadIdx.x;
    const uint32_t gen = (uint32_t)(it + 1), grow = (uint32_t)(a.row0 + row);
    const double *__restrict__ cur = (it & 1) ? a.buf1 : a.buf0;
    double *__restrict__ nxt = (it & 1) ? a.buf0 : a.buf1;
    const double fold = a.fit[row];
    const double *__restrict__ xi = cur + row * ld;
    double *__restrict__ xo = nxt + row * ld;
    const int strategy = STRAT >= 0 ? STRAT : a.strategy, k = donors_of(strategy);
    const bool repair = STRAT >= 0 ? false : a.constraints != 0;
    const bool use_best = strategy == SX_DE_BEST1BIN || strategy == SX_DE_BEST2BIN;
    int64_t d[kMaxDonors];
    int irand;
    if (RNG == SX_RNG_PHILOX) {
        philox_donors(P, k, row, grow, gen, a.key0, a.key1, n, d, irand);
    } else {
#pragma unroll
        for (int t = 0; t < kMaxDonors; ++t) d[t] = t < k ? (int64_t)a.donors[(int64_t)t * P + row] : 0;
        irand = a.irand[row];
    }
    const double *pd[kMaxDonors];
#pragma unroll
    for (int t = 0; t < kMaxDonors; ++t) pd[t] = cur + d[t] * ld;
    const double *__restrict__ gb = a.gbest != nullptr ? a.gbest : cur + sin->gbidx * ld;
    const double F = a.F, CR = a.CR;
    const double *r1row = RNG == SX_RNG_HOST ? a.r1 + row * (int64_t)n : nullptr;
    const double *rsrow = (RNG == SX_RNG_HOST && repair) ? a.resample + row * (int64_t)n : nullptr;

    // thread -> (k256, l): the four elements 256 k256 + l + 64 t, t = 0..3 -- steps q = 4 k256 + t of lane l in the
    // whole-wave layout, i.e. ONE Philox call (slot (q >> 2) * 64 + l = 64 k256 + l) for their crossover uniforms.
    // The per-strategy kernels (two or three donor rows, no bounds) have the four elements' loads in flight together; the
    // generic one (up to five donor rows, bounds, resample draws: 36 doubles per four elements) takes them two at a time --
    // 114-157 VGPRs otherwise, one 512-thread workgroup per CU.
    constexpr int NT = STRAT >= 0 ? 4 : 2;
    struct DeLoads {
        double x[NT], dv[kMaxDonors][NT], gv[NT], r[NT], rs[NT], lo[NT], hi[NT];
    };
    auto de_issue = [&](int g, int t0, int e0, int e1s, DeLoads &L) {
        const int eb = (g >> 6) * 256 + (g & 63);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int e = eb + 64 * (t0 + t);
            const bool in = e >= e0 && e < e1s;
            L.x[t] = in ? xi[e] : 0.0;
            L.gv[t] = (use_best && in) ? gb[e] : 0.0;
#pragma unroll
            for (int s = 0; s < kMaxDonors; ++s) L.dv[s][t] = (s < k && in) ? pd[s][e] : 0.0;
            L.r[t] = 2.0, L.rs[t] = 0.0, L.lo[t] = 0.0, L.hi[t] = 0.0;
            if (RNG == SX_RNG_HOST && in) {
                L.r[t] = r1row[e];
                if (repair) L.rs[t] = rsrow[e];
            }
            if (repair && in) L.lo[t] = a.lower[e], L.hi[t] = a.upper[e];
        }
    };
    auto de_consume = [&](int g, int t0, int e0, int e1, int e1s, DeLoads &L, double *Sd) {
        const int eb = (g >> 6) * 256 + (g & 63);
        if (RNG == SX_RNG_PHILOX) {  // 53-bit crossover uniforms: slot = (q >> 1) * 64 + l, half = q & 1 (q = 4 (g >> 6) + t)
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                const U4 w = philox4x32_10((uint32_t)(((g >> 6) * 4 + t0 + t) >> 1) * 64u + (uint32_t)(g & 63), grow, gen,
                                           kPurposeDeCross, a.key0, a.key1);
                L.r[t] = u53(w.x, w.y);
                L.r[t + 1] = u53(w.z, w.w);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int e = eb + 64 * (t0 + t);
            if (!(e >= e0 && e < e1s)) continue;
            const double v = de_mutant(strategy, L.gv[t], L.dv[0][t], L.dv[1][t], L.dv[2][t], L.dv[3][t], L.dv[4][t], F);
            double cand = (e == irand || L.r[t] <= CR) ? v : L.x[t];  // de/_de.py:341-344
            if (repair && (cand < L.lo[t] || cand > L.hi[t]))         // de/_constraints.py:21-26
                cand = RNG == SX_RNG_HOST ? L.rs[t]
                                          : L.lo[t] + (L.hi[t] - L.lo[t]) *
                                                philox_u53(e, kWave, grow, gen, kPurposeDeResample, a.key0, a.key1);
            stage_put<Obj<FUN>::NEXT>(Sd, e, e0, cand);
            if (!resident && e < e1) xo[e] = cand;  // streamed: the trial goes out as it is produced
        }
    };
    const double fc = wide_row<FUN, T>(c, n, chunk_leaves, S, LA, LB, [&](int e0, int e1, int e1s, double *Sd) {
        for (int g = (e0 >> 8) * 64 + tid; g < ((e1s + 255) >> 8) * 64; g += T) {
#pragma unroll
            for (int t0 = 0; t0 < 4; t0 += NT) {
                DeLoads A;
                de_issue(g, t0, e0, e1s, A);
                de_consume(g, t0, e0, e1, e1s, A, Sd);
            }
        }
    });
    const bool better = fc < fold;  // _common.py:127 strict <
    if (resident || !better) {      // the row of the next generation: the trial (from LDS) or the old row
        for (int eb = tid; eb < n; eb += 8 * T) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = eb + u * T < n ? (better ? S[stage_pos(eb + u * T)] : xi[eb + u * T]) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (eb + u * T < n) xo[eb + u * T] = v[u];
        }
    }
    if (tid == 0) {
        if (better) a.fit[row] = fc;
        if (a.candfit != nullptr) a.candfit[row] = fc;
        a.part_f[row] = better ? fc : fold;
        a.part_i[row] = row;
    }
}

// ---------------------------------------------------------------------------
// PSO / CPSO generation (pso_generation_kernel's general form for wide rows): X, V, pbest in place; pending
// restarts re-seeded here; Shrink takes the row-wide beta from a pass of its own and forms the velocities again.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wide_sort_key(double f) {  // (sx_pso.hip sort_key)
    const unsigned long long b = (unsigned long long)__double_as_longlong(f);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <int FUN, int RNG>
__global__ __launch_bounds__(kGenThreads) void wide_pso_kernel(const sx_pso_args a, const int32_t *__restrict__ plan,
                                                               const int chunk_leaves) {
    constexpr int T = kGenThreads;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s_beta[T / kWave];
    const sx_state *st = a.state;
    if (st->done) return;
    const WideCtx c = wide_ctx(plan);
    const int n = a.n;
    const bool resident = chunk_leaves > 0;
    double *S = lds, *LA = lds + (resident ? wide_resident_elems(n) : kStageElems), *LB = LA + wide_leaf_cap(n);
    const int64_t ld = a.ld, row = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t gen = (uint32_t)(st->it + 1), grow = (uint32_t)(a.row0 + row);
    double fold = a.pbestfit[row];
    bool reseed = false;
    if (RNG == SX_RNG_PHILOX && a.pending_restart != nullptr)
        reseed = a.pending_restart[0] != 0ull && wide_sort_key(fold) >= a.pending_restart[1];
    if (reseed) fold = 1.0e30;
    double *__restrict__ xr = a.X + row * ld;
    double *__restrict__ vr = a.V + row * ld;
    double *__restrict__ pb = a.pbest + row * ld;
    const double *__restrict__ gb = a.gbest;
    const double w = a.w, c1 = a.c1, c2 = a.c2;
    const bool shrink = a.constraints != 0;
    const double *r1row = RNG == SX_RNG_HOST ? a.r1 + row * (int64_t)n : nullptr;
    const double *r2row = RNG == SX_RNG_HOST ? a.r2 + row * (int64_t)n : nullptr;

    // the four elements 256 k256 + l + 64 t of group g = 64 k256 + l (steps q = 4 k256 + t of lane l): position, raw new
    // velocity (cpso/_cpso.py:326) -- two Philox calls (slot (q >> 1) * 64 + l: words (0,1) / (2,3) = (r1, r2) of even /
    // odd q); a re-seeded row draws its position instead of loading it (pso_restart_apply_kernel's draws), V = 0, pbest = X
    struct PsoLoads {
        double x[4], v[4], p[4], gv[4], r1[4], r2[4];
    };
    auto pso_issue = [&](int g, int lo_e, int hi_e, PsoLoads &L) {
        const int eb = (g >> 6) * 256 + (g & 63);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = eb + 64 * t;
            const bool in = e >= lo_e && e < hi_e;
            const bool ldrow = in && !reseed;
            L.x[t] = ldrow ? xr[e] : 0.0;
            L.v[t] = ldrow ? vr[e] : 0.0;
            L.p[t] = ldrow ? pb[e] : 0.0;
            L.gv[t] = in ? gb[e] : 0.0;
            L.r1[t] = (RNG == SX_RNG_HOST && in) ? r1row[e] : 0.0;
            L.r2[t] = (RNG == SX_RNG_HOST && in) ? r2row[e] : 0.0;
        }
    };
    auto pso_form = [&](int g, int lo_e, int hi_e, PsoLoads &L, double(&x)[4], double(&vn)[4], bool(&in)[4]) {
        const int eb = (g >> 6) * 256 + (g & 63);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = eb + 64 * t;
            in[t] = e >= lo_e && e < hi_e;
            x[t] = L.x[t];
        }
        if (RNG == SX_RNG_PHILOX) {
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                const uint32_t slot = (uint32_t)(((g >> 6) * 4 + t) >> 1) * 64u + (uint32_t)(g & 63);
                const U4 wa = philox4x32_10(slot, grow, gen, kPurposePsoR1, a.key0, a.key1);
                const U4 wb = philox4x32_10(slot, grow, gen, kPurposePsoR2, a.key0, a.key1);
                L.r1[t] = u53(wa.x, wa.y), L.r1[t + 1] = u53(wa.z, wa.w), L.r2[t] = u53(wb.x, wb.y), L.r2[t + 1] = u53(wb.z, wb.w);
                if (reseed) {
                    const U4 wr = philox4x32_10(slot, grow, gen - 1u, kPurposePsoRestart, a.key0, a.key1);
                    const double u[2] = {u53(wr.x, wr.y), u53(wr.z, wr.w)};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = eb + 64 * (t + h);
                        if (in[t + h]) {
                            const double lo = a.lower[e];
                            x[t + h] = lo + (a.upper[e] - lo) * u[h];
                            L.p[t + h] = x[t + h];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) vn[t] = pso_velocity(w, L.v[t], c1, L.r1[t], L.p[t], x[t], c2, L.r2[t], L.gv[t]);
    };
    auto elems = [&](int g, int lo_e, int hi_e, double(&x)[4], double(&vn)[4], bool(&in)[4]) {
        PsoLoads L;
        pso_issue(g, lo_e, hi_e, L);
        pso_form(g, lo_e, hi_e, L, x, vn, in);
    };

    double beta = 1.0;
    if (shrink) {  // cpso/_constraints.py:22-50: beta = min over the violated dimensions of (bound - x) / v
        double bmin = __builtin_huge_val();
        for (int g = tid; g < ((n + 255) >> 8) * 64; g += T) {
            double x[4], vn[4];
            bool in[4];
            elems(g, 0, n, x, vn, in);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!in[t]) continue;
                const int e = (g >> 6) * 256 + (g & 63) + 64 * t;
                const double xc = x[t] + vn[t], lo = a.lower[e], hi = a.upper[e];
                if (xc < lo) bmin = fmin(bmin, (lo - x[t]) / vn[t]);
                if (xc > hi) bmin = fmin(bmin, (hi - x[t]) / vn[t]);
            }
        }
        bmin = wave_min_f64(bmin);
        if ((tid & 63) == 0) s_beta[tid >> 6] = bmin;
        __syncthreads();
        bmin = s_beta[0];
        for (int wv = 1; wv < T / kWave; ++wv) bmin = fmin(bmin, s_beta[wv]);
        beta = bmin == __builtin_huge_val() ? 1.0 : bmin;
    }
    auto pso_commit = [&](int g, int e0, int e1, int e1s, PsoLoads &L, double *Sd) {
        double x[4], vn[4];
        bool in[4];
        pso_form(g, e0, e1s, L, x, vn, in);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (!in[t]) continue;
            const int e = (g >> 6) * 256 + (g & 63) + 64 * t;
            const double vf = shrink ? vn[t] * beta : vn[t];  // V *= beta[:, None]
            const double xn = x[t] + vf;
            stage_put<Obj<FUN>::NEXT>(Sd, e, e0, xn);
            if (e < e1) {  // (the look-ahead element is committed by its own chunk: X and V are updated in place)
                vr[e] = vf;
                xr[e] = xn;
                if (reseed) pb[e] = x[t];  // pbest = X of the re-seeded row (kept unless the new position beats 1e30)
            }
        }
    };
    const double fc = wide_row<FUN, T>(c, n, chunk_leaves, S, LA, LB, [&](int e0, int e1, int e1s, double *Sd) {
        for (int g = (e0 >> 8) * 64 + tid; g < ((e1s + 255) >> 8) * 64; g += T) {
            PsoLoads A;
            pso_issue(g, e0, e1s, A);
            pso_commit(g, e0, e1, e1s, A, Sd);
        }
    });
    const bool better = fc < fold;  // _common.py:127 strict <
    if (better) {
        if (!resident) {
            __threadfence_block();
            __syncthreads();  // the positions written above, by other threads of this workgroup
        }
        for (int eb = tid; eb < n; eb += 8 * T) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = eb + u * T < n ? (resident ? S[stage_pos(eb + u * T)] : xr[eb + u * T]) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (eb + u * T < n) pb[eb + u * T] = v[u];
        }
    }
    if (tid == 0) {
        if (better)
            a.pbestfit[row] = fc;
        else if (reseed)
            a.pbestfit[row] = 1.0e30;
        if (a.candfit != nullptr) a.candfit[row] = fc;
        a.part_f[row] = better ? fc : fold;
        a.part_i[row] = row;
    }
}


// ---------------------------------------------------------------------------
// VD-CMA candidates of wide models (vdcma/_vdcma.py:236-248 + the objective), one workgroup per candidate: the normals
// never touch memory -- Philox / Box-Muller in the kernel, laid out as cma_normals_kernel lays them out (element pairs
// (128 k + l, 128 k + 64 + l) = the cosine / sine half of call (slot 64 k + l)) -- the row z waits in LDS (resident rows;
// longer ones park it in the y row they are about to write) for t = z . vn, then y = d o (z + coef t vn),
// x = xmean + sigma y are written, the objective of the un-standardised (Penalize: clipped) x is formed from the staged
// row, and t_k = (y / d) . vn -- what the moment sums need of a selected row (:428-444) -- is left per row.
// Rows 0 and 1 of the generation are +-dy when the mean-shift injection is on (:241-247).
// ---------------------------------------------------------------------------
// (A cap at 64 VGPRs -- four workgroups per CU instead of three -- spills: 24 ... 180 bytes per lane.)
template <int FUN, int T>
__global__ __launch_bounds__(T) void wide_vd_candidates_kernel(const sx_vd_args a, const int64_t gen,
                                                                         const int64_t row0, double *__restrict__ ary_out,
                                                                         double *__restrict__ arx_out,
                                                                         double *__restrict__ fit_out,
                                                                         double *__restrict__ tk_out,
                                                                         const int32_t *__restrict__ plan,
                                                                         const int chunk_leaves) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double s_red[T / kWave];
    const sx_cma_state *st = (const sx_cma_state *)a.state;
    if (st->done) return;
    const WideCtx c = wide_ctx(plan);
    const int n = a.n, tid = (int)threadIdx.x;
    const bool resident = chunk_leaves > 0;
    double *S = lds, *LA = lds + (resident ? wide_resident_elems(n) : kStageElems), *LB = LA + wide_leaf_cap(n);
    const int64_t row = blockIdx.x, grow = row0 + row;
    const double sigma = st->sigma, coef = st->reserved[4];
    const bool inj = st->reserved[3] != 0.0 && grow < 2;
    const double sgn = grow == 0 ? 1.0 : -1.0;
    const bool clip = a.pen_ws != nullptr;
    double *__restrict__ yo = ary_out + row * (int64_t)n;
    // (arx_out may be NULL: x = xmean + sigma y is not kept -- whoever needs it later forms it again from y, sx_vd_args.arx)
    const bool keep_x = arx_out != nullptr;
    double *__restrict__ xo = keep_x ? arx_out + row * (int64_t)n : nullptr;
    auto block_sum = [&](double v) {
        v = wave_sum_butterfly(v, tid & 63);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = v;
        __syncthreads();
        double r = s_red[0];
        for (int w = 1; w < T / kWave; ++w) r += s_red[w];
        return r;
    };
    double t = 0.0;
    if (!inj) {  // the normals of the row and t = z . vn
        double tacc = 0.0;
        for (int g = tid; g < ((n + 255) >> 8) * 64; g += T) {
            const int eb = (g >> 6) * 256 + (g & 63);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e0 = eb + 128 * h, e1 = e0 + 64;
                if (e0 >= n) continue;
                const U4 w = philox4x32_10((uint32_t)((g >> 6) * 2 + h) * 64u + (uint32_t)(g & 63), (uint32_t)grow,
                                           (uint32_t)gen, kPurposeCmaNormal, a.key0, a.key1);
                const double d0 = u53(w.x, w.y), d1 = u53(w.z, w.w);
                const double rad = sqrt(-2.0 * log(1.0 - d0));
                double sn, cs;
                sincos_mid(6.283185307179586 * d1, sn, cs);
                const double z0 = rad * cs, z1 = rad * sn;
                (resident ? S : yo)[resident ? stage_pos(e0) : e0] = z0;
                tacc += z0 * a.vn[e0];
                if (e1 < n) {
                    (resident ? S : yo)[resident ? stage_pos(e1) : e1] = z1;
                    tacc += z1 * a.vn[e1];
                }
            }
        }
        t = block_sum(tacc);  // (its barriers also order the z writes before the reads below)
    }
    double tkacc = 0.0;
    const double fc = wide_row<FUN, T>(c, n, chunk_leaves, S, LA, LB, [&](int e0, int e1, int e1s, double *Sd) {
        for (int g = (e0 >> 8) * 64 + tid; g < ((e1s + 255) >> 8) * 64; g += T) {
            const int eb = (g >> 6) * 256 + (g & 63);
            // (z: the row's normal -- or, in the injected pair's rows, dy: one array; the un-standardisation's xstd / xm ride in
            //  the same batch of loads: a second dependent trip per batch otherwise, 8 per row of 16 384 elements)
            double z[4], vv[4], dd[4], xm0[4], xs[4], xo0[4];
            bool in[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + 64 * u;
                in[u] = e >= e0 && e < e1s;
                z[u] = in[u] ? (inj ? a.dy[e] : (resident ? Sd[stage_pos(e)] : yo[e])) : 0.0;
                vv[u] = in[u] ? a.vn[e] : 0.0;
                dd[u] = in[u] ? a.dvec[e] : 1.0;
                xm0[u] = in[u] ? a.xmean[e] : 0.0;
                xs[u] = in[u] ? a.xstd[e] : 0.0;
                xo0[u] = in[u] ? a.xm[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + 64 * u;
                if (!in[u]) continue;
                const double yd = z[u] + coef * (t * vv[u]);  // y / d of an ordinary row
                const double y = inj ? sgn * z[u] : dd[u] * yd;
                const double x = xm0[u] + sigma * y;
                if (e < e1) {  // (the look-ahead element is written -- and its z read -- by its own chunk)
                    yo[e] = y;
                    if (keep_x) xo[e] = x;
                    tkacc += (inj ? y / dd[u] : yd) * vv[u];  // (the division only for the injected pair: rounding apart the same)
                }
                const double xc = clip ? fmin(fmax(x, -1.0), 1.0) : x;  // cmaes/_constraints.py:29-31
                stage_put<Obj<FUN>::NEXT>(Sd, e, e0, xc * xs[u] + xo0[u]);  // cmaes/_cmaes.py:171
            }
        }
    });
    const double tk = block_sum(tkacc);
    if (tid == 0) {
        fit_out[row] = fc;
        if (tk_out != nullptr) tk_out[row] = tk;
    }
}
template <template <int> class K>
void *pick_fun(int fun_id) {
    switch (fun_id) {
        case SX_FUN_ACKLEY: return K<SX_FUN_ACKLEY>::ptr();
        case SX_FUN_GRIEWANK: return K<SX_FUN_GRIEWANK>::ptr();
        case SX_FUN_QUARTIC: return K<SX_FUN_QUARTIC>::ptr();
        case SX_FUN_RASTRIGIN: return K<SX_FUN_RASTRIGIN>::ptr();
        case SX_FUN_ROSENBROCK: return K<SX_FUN_ROSENBROCK>::ptr();
        case SX_FUN_SPHERE: return K<SX_FUN_SPHERE>::ptr();
        case SX_FUN_STYBLINSKI_TANG: return K<SX_FUN_STYBLINSKI_TANG>::ptr();
    }
    return nullptr;
}
template <template <int> class K>
void *pick_fun_hot(int fun_id) {  // (specialised forms: the four hot objectives, sx_device.hpp hot_objective)
    switch (fun_id) {
        case SX_FUN_ACKLEY: return K<SX_FUN_ACKLEY>::ptr();
        case SX_FUN_RASTRIGIN: return K<SX_FUN_RASTRIGIN>::ptr();
        case SX_FUN_ROSENBROCK: return K<SX_FUN_ROSENBROCK>::ptr();
        case SX_FUN_SPHERE: return K<SX_FUN_SPHERE>::ptr();
    }
    return nullptr;
}
template <int FUN> struct EvalK { static void *ptr() { return (void *)wide_eval_kernel<FUN, false>; } };
template <int FUN> struct EvalPipeK {
    static void *ptr() { return (void *)wide_eval_kernel<FUN, light_objective<FUN>()>; }  // (heavy: the plain form again)
};
template <int FUN> struct DePhK { static void *ptr() { return (void *)wide_de_kernel<FUN, SX_RNG_PHILOX>; } };
template <int FUN> struct DeHoK { static void *ptr() { return (void *)wide_de_kernel<FUN, SX_RNG_HOST>; } };
template <int FUN> struct DePhBestK { static void *ptr() { return (void *)wide_de_kernel<FUN, SX_RNG_PHILOX, SX_DE_BEST1BIN>; } };
template <int FUN> struct DePhRandK { static void *ptr() { return (void *)wide_de_kernel<FUN, SX_RNG_PHILOX, SX_DE_RAND1BIN>; } };
template <int FUN> struct PsoPhK { static void *ptr() { return (void *)wide_pso_kernel<FUN, SX_RNG_PHILOX>; } };
template <int FUN> struct VdCandK { static void *ptr() { return (void *)wide_vd_candidates_kernel<FUN, kGenThreads>; } };
template <int FUN> struct PsoHoK { static void *ptr() { return (void *)wide_pso_kernel<FUN, SX_RNG_HOST>; } };
void *pick_de(const sx_de_args *a) {
    const bool ph = a->rng == SX_RNG_PHILOX;  // (host draws: the generation waits for the host's streams anyway)
    const bool hot = hot_objective(a->fun_id);
    if (hot && ph && a->constraints == 0 && a->strategy == SX_DE_BEST1BIN) return pick_fun_hot<DePhBestK>(a->fun_id);
    if (hot && ph && a->constraints == 0 && a->strategy == SX_DE_RAND1BIN) return pick_fun_hot<DePhRandK>(a->fun_id);
    return ph ? pick_fun<DePhK>(a->fun_id) : pick_fun<DeHoK>(a->fun_id);
}
void *pick_pso(const sx_pso_args *a) { return a->rng == SX_RNG_PHILOX ? pick_fun<PsoPhK>(a->fun_id) : pick_fun<PsoHoK>(a->fun_id); }

std::mutex g_attr_mutex;
std::map<std::pair<int, void *>, size_t> g_attr;  // (device, kernel) -> the dynamic LDS limit it has been given
// A launch with more than 64 KB of dynamic LDS must have been allowed that much (hipFuncAttributeMaxDynamicSharedMemorySize):
// resident rows take up to kResidentLds, streamed rows (4240 + 2 (n / 64 + 2)) doubles -- 99 KB at n = 262144 (the
// limit was always set to kResidentLds, which only worked because this runtime does not enforce it).
int allow_lds(void *fn, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    int dev = 0;
    SX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    auto it = g_attr.find({dev, fn});
    if (it != g_attr.end() && it->second >= bytes) return 0;
    int cap = 0;
    SX_HIP(hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    SX_REQUIRE(cap <= 0 || bytes <= (size_t)cap, "wide rows: the row's LDS stage exceeds the device's LDS per workgroup");
    const size_t want = bytes > kResidentLds ? bytes : kResidentLds;
    SX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    g_attr[{dev, fn}] = want;
    return 0;
}

}  // namespace
namespace sx {
// Build (and cache) the summation plan of this row length now, unless a capture is running: a later wide launch inside a
// stream capture then finds it (sx_eval's eight-lanes-per-row route for rows of 2049 ... 4096 elements never
// built it, and the first captured wide generation failed, silently switching graph replay off for the run).
int wide_warm_plan(int fun_id, int n, hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
    CachedPlan cp;
    return get_plan(sx_fun_terms(fun_id, n), s, &cp);
}
}  // namespace sx
namespace {
struct GenLaunch {
    void *fn;
    const int32_t *plan;
    int chunk_leaves;
    size_t lds;
};
int gen_launch_for(void *fn, int fun_id, int n, hipStream_t s, GenLaunch *out) {
    SX_REQUIRE(fn != nullptr, "wide rows: unknown objective");
    SX_REQUIRE(n <= kWideMaxDim, "dimension above the wide-row limit (n <= 262144)");
    CachedPlan cp;
    if (int rc = get_plan(sx_fun_terms(fun_id, n), s, &cp)) return rc;
    const bool resident = wide_resident(n);
    out->fn = fn;
    out->plan = cp.dev;
    out->chunk_leaves = resident ? cp.nleaf : 0;  // 0: streamed, chunk by chunk (the plan's chunk table)
    out->lds = wide_lds_bytes(n, resident);
    return allow_lds(fn, out->lds);
}

int add_node(hipGraph_t graph, hipGraphNode_t *prev, void *func, dim3 grid, dim3 block, unsigned lds, void **kargs) {
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

}  // namespace

namespace sx {

int wide_eval(int fun_id, const double *X, int64_t P, int n, int64_t ldx, const double *xm, const double *xstd, double *f,
              double *part_f, int64_t *part_i, int clip, const double *pen_v, double *pen_out, hipStream_t s) {
    SX_REQUIRE(n <= kWideMaxDim, "dimension above the wide-row limit (n <= 262144)");
    void *fn = n > SX_WIDE_EVAL_PIPE_FROM ? pick_fun<EvalPipeK>(fun_id) : pick_fun<EvalK>(fun_id);
    SX_REQUIRE(fn != nullptr, "wide rows: unknown objective");
    CachedPlan cp;
    if (int rc = get_plan(sx_fun_terms(fun_id, n), s, &cp)) return rc;
    const size_t lds = wide_lds_bytes(n, false);
    if (int rc = allow_lds(fn, lds)) return rc;
    const int32_t *plan = cp.dev;
    void *kargs[] = {&X, &P, &n, &ldx, &xm, &xstd, &f, &plan, &part_f, &part_i, &clip, &pen_v, &pen_out};
    SX_HIP(hipLaunchKernel(fn, dim3((unsigned)P), dim3(kEvalThreads), kargs, lds, s));
    return 0;
}

int wide_de_launch(const sx_de_args *a, hipStream_t s) {
    GenLaunch g;
    if (int rc = gen_launch_for(pick_de(a), a->fun_id, a->n, s, &g)) return rc;
    sx_de_args args = *a;
    void *kargs[] = {&args, &g.plan, &g.chunk_leaves};
    SX_HIP(hipLaunchKernel(g.fn, dim3((unsigned)a->P), dim3(kGenThreads), kargs, g.lds, s));
    return 0;
}

int wide_de_add_node(hipGraph_t graph, hipGraphNode_t *prev, const sx_de_args *a) {
    GenLaunch g;
    if (int rc = gen_launch_for(pick_de(a), a->fun_id, a->n, nullptr, &g)) return rc;
    sx_de_args args = *a;
    void *kargs[] = {&args, &g.plan, &g.chunk_leaves};
    return add_node(graph, prev, g.fn, dim3((unsigned)a->P), dim3(kGenThreads), (unsigned)g.lds, kargs);
}

int wide_pso_launch(const sx_pso_args *a, hipStream_t s) {
    GenLaunch g;
    if (int rc = gen_launch_for(pick_pso(a), a->fun_id, a->n, s, &g)) return rc;
    sx_pso_args args = *a;
    void *kargs[] = {&args, &g.plan, &g.chunk_leaves};
    SX_HIP(hipLaunchKernel(g.fn, dim3((unsigned)a->P), dim3(kGenThreads), kargs, g.lds, s));
    return 0;
}

int wide_vd_candidates(const sx_vd_args *a, int64_t gen, int64_t row0, int64_t rows, double *ary_out, double *arx_out,
                       double *fit_out, double *tk_out, hipStream_t s) {
    // (Workgroups of 256 or 384 threads -- four per CU instead of three, a generation of ~1000 candidates resident at once --
    // were measured and lose: 273 / 270 against 255-260 us per generation at n = 16 384, P = 1024; 658 / 620 against 570 at
    // n = 65 536, P = 512: profiles/r5_vd_threads.txt.)
    GenLaunch g;
    if (int rc = gen_launch_for(pick_fun<VdCandK>(a->fun_id), a->fun_id, a->n, s, &g)) return rc;
    // Always STREAMED (z parked in the y row): a resident row's 128 KB of LDS leave ONE workgroup per CU, and the generator's
    // arithmetic (Philox, log, sincos: ~65 us of VALU time chip-wide per 1 024 x 16 384 candidates), the row stores and
    // the objective of a candidate then run one after the other -- 216 us per generation at n = 16 384, P = 1 024 against
    // SX_VD_RESIDENT=0's figure in profiles/r5_vd_wide.txt.  (z in registers -- 8 items per thread -- spills: 256 VGPRs + 1.5 KB.)
    static const bool resident_ok = getenv("SX_VD_RESIDENT") != nullptr && getenv("SX_VD_RESIDENT")[0] == '1';
    if (!resident_ok) g.chunk_leaves = 0, g.lds = wide_lds_bytes(a->n, false);
    sx_vd_args args = *a;
    void *kargs[] = {&args, &gen, &row0, &ary_out, &arx_out, &fit_out, &tk_out, &g.plan, &g.chunk_leaves};
    SX_HIP(hipLaunchKernel(g.fn, dim3((unsigned)rows), dim3(kGenThreads), kargs, g.lds, s));
    return 0;
}

int wide_pso_add_node(hipGraph_t graph, hipGraphNode_t *prev, const sx_pso_args *a) {
    GenLaunch g;
    if (int rc = gen_launch_for(pick_pso(a), a->fun_id, a->n, nullptr, &g)) return rc;
    sx_pso_args args = *a;
    void *kargs[] = {&args, &g.plan, &g.chunk_leaves};
    return add_node(graph, prev, g.fn, dim3((unsigned)a->P), dim3(kGenThreads), (unsigned)g.lds, kargs);
}

}  // namespace sx
