// Device-side building blocks shared by the generation kernels (gfx950, wave64).
//
// Thread layout: an individual (row) is owned by LPR = 16, 32 or 64 adjacent lanes of a
// wavefront (lanes_per_row(n): 16 up to n = 64, 32 up to 128, else the whole wave), so a
// wave carries 4, 2 or 1 rows.  Lane l of a row holds elements e = LPR*q + l: every row
// access is a coalesced run of >= 128 bytes, and the fixed per-wave work (state, donors,
// Philox calls, reduction control) is shared by up to 4 rows.  The trial vector U and the
// per-element objective terms are staged in LDS; the row sum is then taken from LDS in
// numpy's pairwise add.reduce order (8 running accumulators over blocks of 8, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), recursion above 128 terms): lane
// l&7 of the row walks chain l&7, the 8-lane tree is three DPP steps.  Fitness values
// therefore reproduce the reference's `.sum()` bit for bit for +,-,* objectives.  The leaf
// table of the recursion travels in the kernel arguments (scalar loads, uniform control flow).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/stochopy_hip.h"

namespace sx {

constexpr int kGroup = 8;     // numpy's 8 running accumulators
constexpr int kWave = 64;     // gfx950 wavefront = one individual
constexpr int kMaxRowsPerBlock = 16;
constexpr int kMaxWavesPerBlock = 8;
constexpr int kMaxLeaf = 64;  // leaves carried in kernel arguments; also one lane per leaf in the merge (<= 64)
constexpr int kMaxDim = 4096; // leaves hold > 64 terms, so n <= 4096 has at most 64; LDS: n+8+2(n/64+2) doubles per row
// Rows of more than kWideFrom elements take the one-workgroup-per-row kernels (sx_wide.hip).  kMaxDim is what the wavefront-
// per-row kernels CAN serve (and what the ordered sweeps, the chained / peer-exchange kernels and full CMA-ES are limited
// to); kWideFrom is where the wide kernels become the faster ones (round 5: first 2560, profiles/r5_wide_threshold.txt; after the
// wide kernels' second pass -- fewer vector instructions, more workgroups per CU -- 2048, profiles/r5_wide_threshold2.txt: at
// n = 2049 sx_eval 0.41 -> 0.47 of the HBM peak, DE 114 -> 99 us, PSO 199 -> 187 us per generation; n = 2048 keeps its compile-time
// plan (sx_eval 0.69 against 0.39), rows of 1792 are a draw).
#ifndef SX_WIDE_FROM
#define SX_WIDE_FROM 2048
#endif
constexpr int kWideFrom = SX_WIDE_FROM;
static_assert(kWideFrom >= 256 && kWideFrom <= kMaxDim, "the wide kernels take over somewhere inside the narrow kernels' range");
// The threshold in effect (host side): kWideFrom unless a run needs the wavefront-per-row kernels for longer rows -- the peer
// exchange and the global-donor gathers live in the chained DE kernel, which serves rows of up to kMaxDim elements
// (sx_set_wide_from; optimize/_de.py raises it to kMaxDim for exchange="p2p" / donors="global" and restores it afterwards).
extern int g_wide_from;
inline int wide_from() { return g_wide_from; }

// Rows of more than 256 elements form their objective terms inside the reduction (row_reduce_leaves_fused:
// one leaf of <= 128 terms per 8-lane group, so >= 3 of the 8 groups are busy); shorter rows have too few
// leaves for that and stage the terms, computed by all lanes, in LDS first.
#ifndef SX_FUSED_ABOVE
#define SX_FUSED_ABOVE 256  // (a build-time knob for A/B measurements: tools/ab_fused.sh)
#endif
__host__ __device__ inline bool fused_terms(int n) { return n > SX_FUSED_ABOVE; }
// doubles of LDS per row.  staged: U[n+8] | A[n] | B[n] | stack[24] | leaf sums [2][n/64+2];
// fused: U[n+8] | leaf sums [2][n/64+2]
__host__ __device__ inline int leaf_cap(int n) { return n / 64 + 2; }
__host__ __device__ inline int lds_row_stride(int n) {
    return fused_terms(n) ? n + 8 + 2 * leaf_cap(n) : 3 * n + 8 + 24 + 2 * leaf_cap(n);
}
// the DE / PSO generation kernels' and the objective kernel's rows: up to 256 elements the objective is a register chain over the staged vector alone
// (row_objective_chain / row_objective_chain_rt), so a row needs n + 8 doubles instead of the term arrays' 3n + ... -- at the
// metric's row length 17 KB per workgroup instead of 54 KB, which with <= 80 VGPRs is three resident workgroups per CU
// instead of two (round 5: what bounds that kernel at large P is latency, profiles/r5_de_gather_probe.txt)
__host__ __device__ inline int gen_row_stride(int n) { return n <= 256 ? n + 8 : lds_row_stride(n); }
// lanes that own one row
// (the smallest of 16/32/64 that covers the row in one batch of 4 steps, else the whole wave)
__host__ __device__ inline int lanes_per_row(int n) { return n <= 64 ? 16 : (n <= 128 ? 32 : 64); }
// waves per workgroup: the largest power of two with <= 16 rows and <= 64 KiB of LDS staging.
// n <= 64 -> 4 waves x 4 rows, 128 -> 8 x 2 (P = 4096 is exactly one workgroup per CU), 256 and 512 -> 8 x 1,
// 1024 -> 4 x 1, 2048 -> 2 x 1.
__host__ __device__ inline int waves_per_block(int n) {
    const int rpw = kWave / lanes_per_row(n);
    const int fit = (64 * 1024) / (8 * lds_row_stride(n) * rpw);
    int w = 1;
    while (2 * w <= fit && 2 * w * rpw <= kMaxRowsPerBlock && 2 * w <= kMaxWavesPerBlock) w *= 2;
    return w;
}
__host__ __device__ inline int rows_per_block(int n) { return waves_per_block(n) * (kWave / lanes_per_row(n)); }

// numpy pairwise-summation plan for an m-term row sum, passed BY VALUE as a kernel argument
struct PlanArg {
    int32_t nleaf;  // 0 when m < 8 (plain left-to-right sum)
    int32_t tail;   // m % 8: terms added one by one after the last leaf's tree
    int32_t mb;     // m / 8
    int32_t depth;  // stack depth of the recursion
    // 32-bit entries: a uniform, run-time-indexed read of a kernel-argument array is then one s_load_dword
    int32_t end[kMaxLeaf];     // end block (exclusive) of leaf t
    int32_t merges[kMaxLeaf];  // stack merges after leaf t
    int32_t mleft[kMaxLeaf];   // merge m adds slot mright[m] into slot mleft[m] (slots = leaf indices;
    int32_t mright[kMaxLeaf];  //  the recursion's combines in order; the total ends in slot 0)
    // work slots of the whole-wave reduction (row_reduce_leaves_fused): a slot is what one 8-lane group carries in one
    // pass of 16 steps -- one leaf, or two consecutive leaves of <= 8 blocks each (numpy cuts a trailing piece of
    // 129..143 terms in two such halves: m = 1023 has 9 leaves, the last two of 8 blocks -- 8 slots, one pass, not two)
    int32_t nslot;
    int32_t sfirst[kMaxLeaf + 1];  // first leaf of slot s; sfirst[nslot] = nleaf
};

// ---------------------------------------------------------------------------
// cross-lane moves inside the 8-lane group (DPP: VALU, no LDS round trip)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: lane i <-> 7-i inside each 8 lanes

__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <bool MUL>
__device__ __forceinline__ double combine(double a, double b) {
    return MUL ? a * b : a + b;
}

// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); + and * commute exactly, so every lane ends with the same bits
template <bool MUL>
__device__ __forceinline__ double group_tree(double r) {
    r = combine<MUL>(r, dpp_f64<kDppXor1>(r));
    r = combine<MUL>(r, dpp_f64<kDppXor2>(r));
    r = combine<MUL>(r, dpp_f64<kDppHalfMirror>(r));  // quads are uniform by now: mirror == xor 4
    return r;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011); counter = (slot,row,gen,purpose), key = seed
// ---------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// numpy legacy double from two words: ((a>>5)*2^26 + (b>>6)) / 2^53
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

enum : uint32_t {
    kPurposeDeCross = 0,
    kPurposeDeDonor = 1,
    kPurposeDeResample = 2,
    kPurposePsoR1 = 3,
    kPurposePsoR2 = 4,
    kPurposePsoRestart = 5,
    kPurposeCmaNormal = 6,
    kPurposeNaUniform = 7,  // NA: the double behind uniform(low, high) of (sample, axis)
    kPurposeInitJitter = 8, // initial population (Philox mode): the uniform inside the stratum, keyed by (row, element)
    kPurposeInitPerm = 9,   // ... and the keys of column j's stratum permutation (slot = j)
};

// Element e of a row sits in lane l = e % LPR of the row's lanes at step q = e / LPR (LPR a power of two).
// 53-bit uniform: slot = (q >> 1) * LPR + l, half = q & 1   (two per call)
__device__ __forceinline__ double philox_u53(int e, int lpr, uint32_t row, uint32_t gen, uint32_t purpose,
                                             uint32_t k0, uint32_t k1) {
    const uint32_t l = (uint32_t)e & (uint32_t)(lpr - 1), q = (uint32_t)e / (uint32_t)lpr;
    const U4 w = philox4x32_10((q >> 1) * (uint32_t)lpr + l, row, gen, purpose, k0, k1);
    return (q & 1u) ? u53(w.z, w.w) : u53(w.x, w.y);
}

// ---------------------------------------------------------------------------
// Objectives: stochopy/factory/benchmark.py:14-156.  term() gives this lane's
// contribution(s) for element e (x = value, xn = value of element e+1).
// ---------------------------------------------------------------------------
constexpr double kTwoPi = 6.283185307179586;  // 2.0 * np.pi

// cos(t) for the arguments the benchmark functions produce (t = 2 pi x, |t| < 1e6; anything else goes to the library).
// Round 4: the library's cosine is ~75 instructions behind a magnitude branch, and the four calls of a lane (four elements
// per batch) run one after the other; this form is ~30 straight-line operations -- Cody-Waite reduction by pi/2 in three
// 33-bit pieces (n * piece is exact for |n| < 2^20, so each step is one rounding), then the classic degree-13 / 14 kernels
// on [-pi/4, pi/4] (the fdlibm coefficients) -- which the scheduler interleaves across the four elements.  Within 1 ulp of
// the library's result for the SAME rounded argument on |x| <= 100 (2e6 random arguments; 2 ulp at |x| ~ 1e4):
// tests/test_gpu_de.py::test_objectives_vs_oracle keeps its 1e-13.  Config 2: 7.04 -> 6.85 us per generation, config 3a 35.3 -> 34.5, 3b 51.7 -> 50.1 (same box, alternating).
#ifndef SX_FAST_COS
#define SX_FAST_COS 2  // 2: cos_2pi below for the 2 pi x terms (round 6), cos_mid elsewhere; 1: cos_mid everywhere (rounds 4-5); 0: the library's cos (A/B)
#endif
__device__ __forceinline__ double cos_mid(double t) {
#if SX_FAST_COS
    const double n = rint(t * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, t);  // first 33 bits of pi/2
    r = fma(-n, 6.07710050630396597660e-11, r);         // next 33
    r = fma(-n, 2.02226624871116645580e-21, r);         // next 33
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                        2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double sn = fma(r * z, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = 1.0 - fma(0.5, z, -(z * pc));
    const int q = (int)n & 3;  // cos(r + n pi/2)
    const double v = (q & 1) ? sn : cs;
    const double fast = (q == 1 || q == 2) ? -v : v;
    return fabs(t) < 1.0e6 ? fast : cos(t);
#else
    return cos(t);
#endif
}

// cos(2 pi x) for the Ackley / Rastrigin terms (benchmark.py:32, :95: np.cos(2.0 * np.pi * x)), round 6.  The period of the
// argument is 1 IN x, so the reduction is exact and free: y = x - rint(x) in [-1/2, 1/2], m = rint(2 y) in {-1, 0, 1},
// r = y - m / 2 in [-1/4, 1/4] (every step exact in binary floating point), cos(2 pi y) = (-1)^m cos(2 pi r), and ONE
// polynomial in z = r^2 -- the Taylor coefficients (-1)^k (2 pi)^(2k) / (2k)!, k <= 11 (truncation 2e-17 at |2 pi r| = pi/2) --
// replaces cos_mid's three Cody-Waite steps, its TWO kernels and the quadrant selects: 17 fp64 operations and 4 integer ones
// against 21 + ~20.  What it returns is cos of the EXACT product 2 pi x where numpy takes the cosine of the ROUNDED one:
// the two differ by |2 pi x| 2^-53 |sin| -- 3.6e-15 on [-5.12, 5.12], 2.3e-14 on Ackley's [-32.768, 32.768] (the size of
// the reference's own argument rounding; absolute error of the polynomial itself ~1.5e-16) -- inside the 1e-13 bar on objective values
// (tests/test_gpu_de.py::test_objectives_vs_oracle, factory_kat.json).  SX_FAST_COS=1: cos_mid(2 pi x) as in rounds 4-5 (A/B).
__device__ __forceinline__ double cos_2pi(double x) {
#if SX_FAST_COS == 2
    const double y = x - rint(x);
    const double m = rint(2.0 * y);
    const double r = fma(m, -0.5, y);
    const double z = r * r;
    double p = -0x1.52ae4120fde27p-12;
    p = fma(p, z, 0x1.ef6e308d6d1c4p-9);
    p = fma(p, z, -0x1.2a0c591af8314p-5);
    p = fma(p, z, 0x1.20c62c2f2d7f5p-2);
    p = fma(p, z, -0x1.b6e24f44b128fp+0);
    p = fma(p, z, 0x1.f9d38a3763cc3p+2);
    p = fma(p, z, -0x1.a6d1f2a204a8cp+4);
    p = fma(p, z, 0x1.e1f506891babbp+5);
    p = fma(p, z, -0x1.55d3c7e3cbffap+6);
    p = fma(p, z, 0x1.03c1f081b5ac4p+6);
    p = fma(p, z, -0x1.3bd3cc9be45dep+4);
    p = fma(p, z, 1.0);
    // (-1)^m: m is -1, 0 or 1 -- its lowest integer bit moved onto the sign
    const unsigned flip = ((unsigned)(int)m & 1u) << 31;
    return __hiloint2double(__double2hiint(p) ^ (int)flip, __double2loint(p));
#else
    return cos_mid(kTwoPi * x);
#endif
}

// sine and cosine of one argument with cos_mid's reduction and kernels (|t| < 1e6; the wide VD-CMA candidates kernel's
// Box-Muller angles 2 pi u, u in [0, 1): the library's sincos is ~200 instructions behind its large-argument branch)
__device__ __forceinline__ void sincos_mid(double t, double &sn_out, double &cs_out) {
    const double n = rint(t * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, t);
    r = fma(-n, 6.07710050630396597660e-11, r);
    r = fma(-n, 2.02226624871116645580e-21, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                        2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double sn = fma(r * z, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                   -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cs = 1.0 - fma(0.5, z, -(z * pc));
    const int q = (int)n & 3;  // sin / cos (r + n pi/2)
    const double c = (q & 1) ? sn : cs, s = (q & 1) ? cs : sn;
    cs_out = (q == 1 || q == 2) ? -c : c;
    sn_out = (q >= 2) ? -s : s;
}

template <int FUN>
struct Obj;
// Round 6 (920 kernel instantiations): the SPECIALISED forms of the generation kernels -- strategy at
// compile time, one-batch rows, PLAIN PSO, compile-time row length of the ordered sweeps -- exist for the four objectives the
// BASELINE configs and the reference's tests use; Griewank, Quartic and Styblinski-Tang take the general form of the same kernel
// (same arithmetic, same results).  Of the DE strategies only best1bin and rand1bin are specialised.
inline bool hot_objective(int fun_id) {
    return fun_id == SX_FUN_ACKLEY || fun_id == SX_FUN_RASTRIGIN || fun_id == SX_FUN_ROSENBROCK || fun_id == SX_FUN_SPHERE;
}

template <>
struct Obj<SX_FUN_ACKLEY> {  // benchmark.py:14-34
    static constexpr bool NEXT = false, BMUL = false, TWO = true;
    static __device__ __forceinline__ void term(double x, double, int, double &a, double &b) {
        a = x * x;
        b = cos_2pi(x);
    }
    static __device__ __forceinline__ double finish(double sa, double sb, int n) {
        const double e = 2.7182818284590451;
        const double inv = 1.0 / (double)n;
        const double s1 = sqrt(inv * sa);
        const double s2 = inv * sb;
        return ((20.0 + e) - 20.0 * exp(-0.2 * s1)) - exp(s2);
    }
};

template <>
struct Obj<SX_FUN_GRIEWANK> {  // benchmark.py:37-56
    static constexpr bool NEXT = false, BMUL = true, TWO = true;
    static __device__ __forceinline__ void term(double x, double, int e, double &a, double &b) {
        a = x * x;
        b = cos(x / sqrt((double)(e + 1)));
    }
    static __device__ __forceinline__ double finish(double sa, double pb, int) {
        return (1.0 + sa / 4000.0) - pb;
    }
};

template <>
struct Obj<SX_FUN_QUARTIC> {  // benchmark.py:59-76
    static constexpr bool NEXT = false, BMUL = false, TWO = false;
    static __device__ __forceinline__ void term(double x, double, int e, double &a, double &b) {
        const double x2 = x * x;
        a = (double)(e + 1) * (x2 * x2);
        b = 0.0;
    }
    static __device__ __forceinline__ double finish(double sa, double, int) { return sa; }
};

template <>
struct Obj<SX_FUN_RASTRIGIN> {  // benchmark.py:79-97
    static constexpr bool NEXT = false, BMUL = false, TWO = false;
    static __device__ __forceinline__ void term(double x, double, int, double &a, double &b) {
        a = x * x - 10.0 * cos_2pi(x);
        b = 0.0;
    }
    static __device__ __forceinline__ double finish(double sa, double, int n) { return 10.0 * (double)n + sa; }
};

template <>
struct Obj<SX_FUN_ROSENBROCK> {  // benchmark.py:100-118
    static constexpr bool NEXT = true, BMUL = false, TWO = true;
    static __device__ __forceinline__ void term(double x, double xn, int, double &a, double &b) {
        const double t = xn - x * x;
        a = t * t;
        const double u = 1.0 - x;
        b = u * u;
    }
    static __device__ __forceinline__ double finish(double sa, double sb, int) { return 100.0 * sa + sb; }
};

template <>
struct Obj<SX_FUN_SPHERE> {  // benchmark.py:121-136
    static constexpr bool NEXT = false, BMUL = false, TWO = false;
    static __device__ __forceinline__ void term(double x, double, int, double &a, double &b) {
        a = x * x;
        b = 0.0;
    }
    static __device__ __forceinline__ double finish(double sa, double, int) { return sa; }
};

template <>
struct Obj<SX_FUN_STYBLINSKI_TANG> {  // benchmark.py:139-156
    static constexpr bool NEXT = false, BMUL = false, TWO = false;
    static __device__ __forceinline__ void term(double x, double, int, double &a, double &b) {
        const double x2 = x * x;
        a = (x2 * x2 - 16.0 * x2) + 5.0 * x;
        b = 0.0;
    }
    static __device__ __forceinline__ double finish(double sa, double, int n) {
        return 0.5 * sa + 39.16599 * (double)n;
    }
};

// ---------------------------------------------------------------------------
// Row sum in numpy order from LDS-staged terms.  Executed by all 64 lanes: the 8
// lane groups compute the same sum redundantly (LDS broadcasts), so every lane
// ends with identical bits and no cross-lane broadcast is needed.
// ---------------------------------------------------------------------------
constexpr int kLeafBlocks = 16;  // a leaf of numpy's recursion has at most 128 terms
constexpr int kStack = 12;       // leaf sums awaiting their sibling; depth <= log2(m/64)+1

// Two value streams at once (A: sum, B: sum or product) so their dependent chains
// interleave; LDS reads are issued 8 blocks ahead of the adds.  S: 2*kStack doubles of
// LDS scratch per wave for the stack of leaf sums (only touched when the row has > 1 leaf).
template <bool TWO, bool BMUL>
__device__ __forceinline__ void row_reduce2(const double *A, const double *B, double *S, const PlanArg &p, int lane,
                                            double &sa, double &sb) {
    const int j = lane & (kGroup - 1);
    const double identB = BMUL ? 1.0 : 0.0;
    // tail terms (after the last leaf's tree; the whole row when m < 8), fetched up front
    double ta[kGroup - 1], tb[kGroup - 1];
    const int t0 = p.mb * kGroup;
#pragma unroll
    for (int k = 0; k < kGroup - 1; ++k) {
        ta[k] = (k < p.tail) ? A[t0 + k] : 0.0;
        tb[k] = (TWO && k < p.tail) ? B[t0 + k] : identB;
    }
    int sp = 0;
    double curA = 0.0, curB = identB;
    int b0 = 0;
    for (int leaf = 0; leaf < p.nleaf; ++leaf) {
        const int b1 = (int)p.end[leaf];
        const int cnt = b1 - b0;
        double chA = 0.0, chB = identB;
#pragma unroll
        for (int h = 0; h < kLeafBlocks; h += 8) {
            double va[8], vb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = h + t < cnt;
                va[t] = in ? A[(b0 + h + t) * kGroup + j] : 0.0;
                vb[t] = (TWO && in) ? B[(b0 + h + t) * kGroup + j] : identB;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (h + t == 0) {
                    chA = va[0];
                    chB = vb[0];
                } else if (h + t < cnt) {  // uniform
                    chA = chA + va[t];
                    if (TWO) chB = combine<BMUL>(chB, vb[t]);
                }
            }
        }
        curA = group_tree<false>(chA);
        if (TWO) curB = group_tree<BMUL>(chB);
        if (leaf == p.nleaf - 1) {
#pragma unroll
            for (int k = 0; k < kGroup - 1; ++k) {
                if (k < p.tail) {
                    curA = curA + ta[k];
                    if (TWO) curB = combine<BMUL>(curB, tb[k]);
                }
            }
        }
        if (p.nleaf > 1) {  // push, then merge with finished siblings; every lane stores the same bits
            S[sp] = curA;
            if (TWO) S[kStack + sp] = curB;
            ++sp;
            for (int merges = (int)p.merges[leaf]; merges > 0; --merges) {
                S[sp - 2] = S[sp - 2] + S[sp - 1];
                if (TWO) S[kStack + sp - 2] = combine<BMUL>(S[kStack + sp - 2], S[kStack + sp - 1]);
                --sp;
            }
        }
        b0 = b1;
    }
    if (p.nleaf == 0) {  // m < 8: plain left-to-right
#pragma unroll
        for (int k = 0; k < kGroup - 1; ++k) {
            if (k < p.tail) {
                curA = curA + ta[k];
                if (TWO) curB = combine<BMUL>(curB, tb[k]);
            }
        }
    }
    if (p.nleaf > 1) {
        curA = S[0];
        if (TWO) curB = S[kStack];
    }
    sa = 0.0 + curA;  // add.reduce starts from the identity
    sb = (TWO && !BMUL) ? 0.0 + curB : curB;
}

// value held by lane `idx` of this lane's row (idx uniform).  A whole-wave row can use v_readlane.
template <int LPR>
__device__ __forceinline__ double row_lane_value(double v, int idx, int l) {
    if (LPR == kWave) return readlane_f64(v, idx);
    return __shfl(v, (int)(threadIdx.x & 63) - l + idx, kWave);
}

// Rows of exactly M = 64, 128 or 256 terms (the PSO kernel's whole-batch rows with an objective that has one term per
// element): numpy's plan is known when the kernel is compiled -- one leaf of M/8 blocks, or two leaves of 16 -- so
// the chain, the tree and the leaf sum are straight-line code instead of loops over the plan arrays (scalar loads,
// selects and bound checks per step).  Same additions in the same order as row_reduce2 / row_reduce_leaves: same bits.
template <bool TWO, bool BMUL, int LPR, int M>
__device__ __forceinline__ void row_reduce_fixed(const double *A, const double *B, int l, double &sa, double &sb) {
    static_assert(M == 64 || M == 128 || M == 256, "one or two full leaves");
    constexpr int NLEAF = M > 128 ? 2 : 1;
    constexpr int BLK = M / NLEAF / kGroup;  // blocks per leaf: 8 or 16
    const int j = l & (kGroup - 1), grp = l >> 3;
    const int leaf = (NLEAF == 2 && grp == 1) ? 1 : 0;  // the other lane groups repeat leaf 0 (LDS broadcasts)
    const double identB = BMUL ? 1.0 : 0.0;
    const double *a = A + leaf * (BLK * kGroup) + j, *b = B + leaf * (BLK * kGroup) + j;
    double chA = 0.0, chB = identB;
#pragma unroll
    for (int h = 0; h < BLK; h += 8) {
        double va[8], vb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            va[t] = a[(h + t) * kGroup];
            vb[t] = TWO ? b[(h + t) * kGroup] : identB;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (h + t == 0) {
                chA = va[0];
                chB = vb[0];
            } else {
                chA = chA + va[t];
                if (TWO) chB = combine<BMUL>(chB, vb[t]);
            }
        }
    }
    double curA = group_tree<false>(chA);
    double curB = TWO ? group_tree<BMUL>(chB) : identB;
    if (NLEAF == 2) {  // leaf 0 lives in lane group 0, leaf 1 in group 1
        curA = row_lane_value<LPR>(curA, 0, l) + row_lane_value<LPR>(curA, kGroup, l);
        if (TWO) curB = combine<BMUL>(row_lane_value<LPR>(curB, 0, l), row_lane_value<LPR>(curB, kGroup, l));
    }
    sa = 0.0 + curA;  // add.reduce starts from the identity
    sb = (TWO && !BMUL) ? 0.0 + curB : curB;
}

// Any compile-time number of terms M >= 8 (whole-batch rows of an objective with n - 1 terms: 63, 127, 255): numpy's
// recursion (loops_utils.h.src pairwise sum: up to 128 terms = eight accumulators over the 8-blocks, the tree, then the
// tail; above, split at n/2 rounded down to a multiple of 8) unrolled by the compiler.  Every 8-lane group walks the
// same chain (LDS broadcasts), the leaves one after the other: a few more chain steps than the leaves-in-parallel form,
// none of its bookkeeping.  Same additions in the same order: same bits.
template <bool TWO, bool BMUL, int OFF, int M>
__device__ __forceinline__ void pairwise_static(const double *A, const double *B, int j, double &ra, double &rb) {
    static_assert(M >= 8, "shorter sums are plain loops");
    if constexpr (M <= 128) {
        constexpr int BLK = M / kGroup, TAIL = M % kGroup;
        double chA = A[OFF + j], chB = TWO ? B[OFF + j] : (BMUL ? 1.0 : 0.0);
#pragma unroll
        for (int h0 = 1; h0 < BLK; h0 += 8) {  // reads run up to 8 blocks ahead of the adds
            double va[8], vb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                va[t] = h0 + t < BLK ? A[OFF + (h0 + t) * kGroup + j] : 0.0;
                vb[t] = (TWO && h0 + t < BLK) ? B[OFF + (h0 + t) * kGroup + j] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (h0 + t < BLK) {
                    chA = chA + va[t];
                    if (TWO) chB = combine<BMUL>(chB, vb[t]);
                }
            }
        }
        double ta[TAIL > 0 ? TAIL : 1], tb[TAIL > 0 ? TAIL : 1];
#pragma unroll
        for (int k = 0; k < TAIL; ++k) {
            ta[k] = A[OFF + BLK * kGroup + k];
            tb[k] = TWO ? B[OFF + BLK * kGroup + k] : 0.0;
        }
        ra = group_tree<false>(chA);
        rb = TWO ? group_tree<BMUL>(chB) : chB;
#pragma unroll
        for (int k = 0; k < TAIL; ++k) {
            ra = ra + ta[k];
            if (TWO) rb = combine<BMUL>(rb, tb[k]);
        }
    } else {
        constexpr int N2 = (M / 2) - ((M / 2) % kGroup);
        double la, lb, qa, qb;
        pairwise_static<TWO, BMUL, OFF, N2>(A, B, j, la, lb);
        pairwise_static<TWO, BMUL, OFF + N2, M - N2>(A, B, j, qa, qb);
        ra = la + qa;
        rb = TWO ? combine<BMUL>(lb, qb) : lb;
    }
}

template <bool TWO, bool BMUL, int M>
__device__ __forceinline__ void row_reduce_static(const double *A, const double *B, int l, double &sa, double &sb) {
    double ra, rb;
    pairwise_static<TWO, BMUL, 0, M>(A, B, l & (kGroup - 1), ra, rb);
    sa = 0.0 + ra;  // add.reduce starts from the identity
    sb = !TWO ? (BMUL ? 1.0 : 0.0) : (BMUL ? rb : 0.0 + rb);
}

// Rows with several leaves (n > 128): the leaves of numpy's recursion are independent, so the LPR/8
// 8-lane groups of the row take one leaf each (chain + tree [+ tail]); the leaf sums meet in LDS and are
// then merged in recursion order.  L: 2*leaf_cap doubles of LDS scratch (leaf sums), S: the merge stack.
template <bool TWO, bool BMUL, int LPR>
__device__ __forceinline__ void row_reduce_leaves(const double *A, const double *B, double *S, double *L, int lcap,
                                                  const PlanArg &p, int l, double &sa, double &sb) {
    constexpr int NG = LPR / kGroup;
    const int j = l & (kGroup - 1), grp = l >> 3;
    const double identB = BMUL ? 1.0 : 0.0;
    const int t0 = p.mb * kGroup;
    for (int leaf0 = 0; leaf0 < p.nleaf; leaf0 += NG) {
        // this group's leaf: block range via uniform (scalar) plan reads + selects
        int b0 = 0, b1 = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int lf = leaf0 + g;
            const int e1 = lf < p.nleaf ? (int)p.end[lf] : 0;
            const int e0 = (lf > 0 && lf <= p.nleaf) ? (int)p.end[lf - 1] : 0;
            if (grp == g) {
                b0 = e0;
                b1 = e1;
            }
        }
        const int leaf = leaf0 + grp;
        const int cnt = b1 - b0;  // 0 for groups beyond the last leaf
        double chA = 0.0, chB = identB;
#pragma unroll
        for (int h = 0; h < kLeafBlocks; h += 8) {
            double va[8], vb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = h + t < cnt;
                va[t] = in ? A[(b0 + h + t) * kGroup + j] : 0.0;
                vb[t] = (TWO && in) ? B[(b0 + h + t) * kGroup + j] : identB;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (h + t == 0) {
                    chA = va[0];
                    chB = vb[0];
                } else if (h + t < cnt) {
                    chA = chA + va[t];
                    if (TWO) chB = combine<BMUL>(chB, vb[t]);
                }
            }
        }
        double curA = group_tree<false>(chA);
        double curB = TWO ? group_tree<BMUL>(chB) : identB;
        if (leaf == p.nleaf - 1) {
            for (int k = 0; k < p.tail; ++k) {
                curA = curA + A[t0 + k];
                if (TWO) curB = combine<BMUL>(curB, B[t0 + k]);
            }
        }
        if (cnt > 0 && j == 0) {
            L[leaf] = curA;
            if (TWO) L[lcap + leaf] = curB;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // merge in recursion order: leaf t's sums sit in lane t; the host-built program (pair[m] = left, right
    // slot) is replayed with one readlane + one add per merge -- no LDS round trips.  Result in slot 0.
    double vA = l < p.nleaf ? L[l] : 0.0;
    double vB = (TWO && l < p.nleaf) ? L[lcap + l] : identB;
    for (int m = 0; m + 1 < p.nleaf; ++m) {
        const int left = (int)p.mleft[m], right = (int)p.mright[m];  // uniform (scalar loads)
        const double rA = row_lane_value<LPR>(vA, right, l);
        const double rB = TWO ? row_lane_value<LPR>(vB, right, l) : identB;
        if (l == left) {
            vA = vA + rA;
            if (TWO) vB = combine<BMUL>(vB, rB);
        }
    }
    sa = 0.0 + row_lane_value<LPR>(vA, 0, l);
    const double rb = TWO ? row_lane_value<LPR>(vB, 0, l) : identB;
    sb = (TWO && !BMUL) ? 0.0 + rb : rb;
}

// The same for whole-wave rows (n > 128), with the objective terms computed on the way: the chain lane that
// adds term e reads U[e] (and U[e+1]) from LDS and forms the term itself -- every lane still handles m/64
// terms, but no term array is written or staged, so a row needs n+8 doubles of LDS instead of 3n+8 (3x the
// rows per CU at n = 1024).  Same operations on the same values: same bits as the staged form.
// PAIRED: some slot of the plan holds two leaves (PlanArg::nslot < nleaf -- only when that saves a pass); otherwise slot s
// is leaf s and nothing of the pairing survives in the code.
#ifndef SX_FUSED_SELECT
#define SX_FUSED_SELECT 0  // A/B switch (round 5): 1 = a select per term instead of a branch per term in the chains (what the
                           // wide kernels do, sx_wide.hip); here measured neutral (DE Rosenbrock n=300 43.8 -> 44.6 us,
                           // n=1500 75.2 -> 74.4: profiles/r5_narrow_ab.txt, which has both switches on): off
#endif
#ifndef SX_FUSED_TAIL_BY_LANE
#define SX_FUSED_TAIL_BY_LANE 1  // A/B switch (round 5): 0 = every lane of the last leaf's group forms every tail term
                                 // (DE Rastrigin n=300 55.3 -> 52.3 us, n=700 44.7 -> 43.5: profiles/r5_narrow_ab.txt)
#endif
template <int FUN, int LPR, bool PAIRED>
__device__ __forceinline__ void row_reduce_leaves_fused_impl(const double *U, double *L, int lcap, int m, const PlanArg &p,
                                                             int l, double &sa, double &sb) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    constexpr int NG = LPR / kGroup;
    constexpr bool kSelectOnChain = SX_FUSED_SELECT && !O::TWO && FUN != SX_FUN_RASTRIGIN;  // one cheap term per element
    const int j = l & (kGroup - 1), grp = l >> 3;
    const double identB = BMUL ? 1.0 : 0.0;
    const int t0 = p.mb * kGroup;
    for (int slot0 = 0; slot0 < p.nslot; slot0 += NG) {
        // this group's slot: leaves lfA (blocks [a0, a1)) and, when it holds two, lfA + 1 (blocks [a1, c1))
        int lfA = 0, nl = 0, a0 = 0, a1 = 0, c1 = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sl = slot0 + g;
            const bool on = sl < p.nslot;
            const int f = !on ? 0 : (PAIRED ? (int)p.sfirst[sl] : sl), f2 = !on ? 0 : (PAIRED ? (int)p.sfirst[sl + 1] : sl + 1);
            const int e0 = (on && f > 0) ? (int)p.end[f - 1] : 0;
            const int e1 = on ? (int)p.end[f] : 0;
            const int e2 = (on && f2 - f == 2) ? (int)p.end[f + 1] : e1;
            if (grp == g) lfA = f, nl = f2 - f, a0 = e0, a1 = e1, c1 = e2;
        }
        const bool two = PAIRED && nl == 2;
        const int cntA = a1 - a0;  // 0 for groups beyond the last slot
        double chA = 0.0, chB = identB;
        // steps 0..7: the first (or only) leaf
        if constexpr (kSelectOnChain) {
            // cheap terms: the blocks a short leaf does not have are read all the same (inside the workgroup's LDS, or
            // past it: zeros) and their terms dropped from the chain -- a select per term instead of a branch per term
            double x[8], xn[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int e = (a0 + t) * kGroup + j;
                x[t] = U[e];
                xn[t] = O::NEXT ? U[e + 1] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = t < cntA;
                double a, b;
                O::term(x[t], xn[t], (a0 + t) * kGroup + j, a, b);
                if (t == 0) {
                    chA = in ? a : 0.0;
                    chB = in ? b : identB;
                } else {
                    chA = in ? chA + a : chA;
                    if (TWO) chB = in ? combine<BMUL>(chB, b) : chB;
                }
            }
        } else {
            double x[8], xn[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool in = t < cntA;
                const int e = (a0 + t) * kGroup + j;
                x[t] = in ? U[e] : 0.0;
                xn[t] = (O::NEXT && in) ? U[e + 1] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < cntA) {
                    double a, b;
                    O::term(x[t], xn[t], (a0 + t) * kGroup + j, a, b);
                    if (t == 0) {
                        chA = a;
                        chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        }
        if (two) {  // the first leaf of a pair (<= 8 blocks) is complete: its tree, its sum; the chains start again
            const double curA = group_tree<false>(chA);
            const double curB = TWO ? group_tree<BMUL>(chB) : identB;
            if (j == 0) {
                L[lfA] = curA;
                if (TWO) L[lcap + lfA] = curB;
            }
        }
        // steps 8..15: the rest of the only leaf, or the second leaf of the pair from its first block
        const int base2 = two ? a1 - 8 : a0, lim2 = two ? 8 + (c1 - a1) : cntA;
        if constexpr (kSelectOnChain) {
            double x[8], xn[8];
#pragma unroll
            for (int t = 8; t < kLeafBlocks; ++t) {
                const int e = (base2 + t) * kGroup + j;
                x[t - 8] = U[e];
                xn[t - 8] = O::NEXT ? U[e + 1] : 0.0;
            }
#pragma unroll
            for (int t = 8; t < kLeafBlocks; ++t) {
                const bool in = t < lim2;
                double a, b;
                O::term(x[t - 8], xn[t - 8], (base2 + t) * kGroup + j, a, b);
                if (t == 8 && two) {
                    chA = in ? a : chA;
                    chB = in ? b : chB;
                } else {
                    chA = in ? chA + a : chA;
                    if (TWO) chB = in ? combine<BMUL>(chB, b) : chB;
                }
            }
        } else {
            double x[8], xn[8];
#pragma unroll
            for (int t = 8; t < kLeafBlocks; ++t) {
                const bool in = t < lim2;
                const int e = (base2 + t) * kGroup + j;
                x[t - 8] = in ? U[e] : 0.0;
                xn[t - 8] = (O::NEXT && in) ? U[e + 1] : 0.0;
            }
#pragma unroll
            for (int t = 8; t < kLeafBlocks; ++t) {
                if (t < lim2) {
                    double a, b;
                    O::term(x[t - 8], xn[t - 8], (base2 + t) * kGroup + j, a, b);
                    if (t == 8 && two) {
                        chA = a;
                        chB = b;
                    } else {
                        chA = chA + a;
                        if (TWO) chB = combine<BMUL>(chB, b);
                    }
                }
            }
        }
        const int leaf = lfA + (two ? 1 : 0);
        double curA = group_tree<false>(chA);
        double curB = TWO ? group_tree<BMUL>(chB) : identB;
        if (leaf == p.nleaf - 1 && p.tail > 0) {
#if SX_FUSED_TAIL_BY_LANE
            // the tail terms: lane k of the group forms term k, the sums take them in order (one term's arithmetic, not
            // `tail` times that, while the other seven groups wait)
            double ta = 0.0, tb = identB;
            if (j < p.tail) O::term(U[t0 + j], O::NEXT ? U[t0 + j + 1] : 0.0, t0 + j, ta, tb);
#pragma unroll
            for (int k = 0; k < kGroup - 1; ++k) {
                const int src = (int)(threadIdx.x & (kWave - kGroup)) + k;  // lane k of this 8-lane group
                const double va = __shfl(ta, src, kWave);
                if (k < p.tail) curA = curA + va;
                if (TWO) {
                    const double vb = __shfl(tb, src, kWave);
                    if (k < p.tail) curB = combine<BMUL>(curB, vb);
                }
            }
#else
            for (int k = 0; k < p.tail; ++k) {
                double a, b;
                O::term(U[t0 + k], O::NEXT ? U[t0 + k + 1] : 0.0, t0 + k, a, b);
                curA = curA + a;
                if (TWO) curB = combine<BMUL>(curB, b);
            }
#endif
        }
        if (nl > 0 && j == 0) {
            L[leaf] = curA;
            if (TWO) L[lcap + leaf] = curB;
        }
    }
    (void)m;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double vA = l < p.nleaf ? L[l] : 0.0;
    double vB = (TWO && l < p.nleaf) ? L[lcap + l] : identB;
    for (int mm = 0; mm + 1 < p.nleaf; ++mm) {
        const int left = (int)p.mleft[mm], right = (int)p.mright[mm];  // uniform (scalar loads)
        const double rA = row_lane_value<LPR>(vA, right, l);
        const double rB = TWO ? row_lane_value<LPR>(vB, right, l) : identB;
        if (l == left) {
            vA = vA + rA;
            if (TWO) vB = combine<BMUL>(vB, rB);
        }
    }
    sa = 0.0 + row_lane_value<LPR>(vA, 0, l);
    const double rb = TWO ? row_lane_value<LPR>(vB, 0, l) : identB;
    sb = (TWO && !BMUL) ? 0.0 + rb : rb;
}
template <int FUN, int LPR>
__device__ __forceinline__ void row_reduce_leaves_fused(const double *U, double *L, int lcap, int m, const PlanArg &p,
                                                        int l, double &sa, double &sb) {
    if (p.nslot != p.nleaf)  // (uniform)
        row_reduce_leaves_fused_impl<FUN, LPR, true>(U, L, lcap, m, p, l, sa, sb);
    else
        row_reduce_leaves_fused_impl<FUN, LPR, false>(U, L, lcap, m, p, l, sa, sb);
}

// ---------------------------------------------------------------------------
// Whole-wave rows whose number of terms M is known when the kernel is compiled (sx_eval's streaming form: n = 512 / 1024 /
// 2048): numpy's plan as constants.  The run-time form above spends more instructions on the plan (scalar loads of the
// leaf table, selects per group, a branch per step, leaf sums through LDS, a merge loop over lane values) than on the
// terms; here a pass is straight-line code: one base address per lane, 16 steps whose LDS offsets are immediates and whose
// bound checks survive only where some leaf of the pass is shorter than the step, leaf sums picked up with v_readlane as
// uniform values, the tail terms formed by seven lanes at once and added in order, the merges as a fixed sequence of
// additions.  Same operations on the same values in the same order: same bits.
// ---------------------------------------------------------------------------
template <int M>
struct LongPlan {
    static constexpr int kCap = M / 64 + 2;
    int nleaf = 0, nmerge = 0, nslot = 0;
    int end[kCap] = {};                      // end block (exclusive) of leaf t
    int mleft[kCap] = {}, mright[kCap] = {}; // the recursion's combines in order: leaf-slot mleft += leaf-slot mright
    int sfirst[kCap + 1] = {};               // first leaf of work slot s (a slot: one leaf, or two of <= 8 blocks each)
};
template <int M>
constexpr int long_plan_rec(LongPlan<M> &p, int lo, int n) {  // loops_utils.h.src pairwise sum; returns the slot of the sum
    if (n <= 128) {
        const int t = p.nleaf++;
        p.end[t] = (lo + n) / kGroup;
        return t;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int a = long_plan_rec(p, lo, n2);
    const int b = long_plan_rec(p, lo + n2, n - n2);
    p.mleft[p.nmerge] = a, p.mright[p.nmerge] = b;
    ++p.nmerge;
    return a;
}
template <int M>
constexpr LongPlan<M> make_long_plan() {
    LongPlan<M> p{};
    (void)long_plan_rec(p, 0, M);
    for (int t = 0; t < p.nleaf;) {
        const int b0 = t > 0 ? p.end[t - 1] : 0, b1 = p.end[t];
        const bool pair = t + 1 < p.nleaf && b1 - b0 <= 8 && p.end[t + 1] - b1 <= 8;
        p.sfirst[p.nslot++] = t;
        t += pair ? 2 : 1;
    }
    if ((p.nslot + 7) / 8 == (p.nleaf + 7) / 8) {  // pairing saves no pass: one leaf per slot
        p.nslot = p.nleaf;
        for (int t = 0; t < p.nleaf; ++t) p.sfirst[t] = t;
    }
    p.sfirst[p.nslot] = p.nleaf;
    return p;
}
template <int M>
struct LongPlanOf {
    static constexpr LongPlan<M> value = make_long_plan<M>();
};
// slot s of the plan: blocks [e0, e1) and, for a pair, [e1, e2)
template <int M>
constexpr int lp_e0(int s) {
    const int f = LongPlanOf<M>::value.sfirst[s];
    return f > 0 ? LongPlanOf<M>::value.end[f - 1] : 0;
}
template <int M>
constexpr int lp_e1(int s) { return LongPlanOf<M>::value.end[LongPlanOf<M>::value.sfirst[s]]; }
template <int M>
constexpr bool lp_two(int s) { return LongPlanOf<M>::value.sfirst[s + 1] - LongPlanOf<M>::value.sfirst[s] == 2; }
template <int M>
constexpr int lp_e2(int s) { return lp_two<M>(s) ? LongPlanOf<M>::value.end[LongPlanOf<M>::value.sfirst[s] + 1] : lp_e1<M>(s); }
// is step t of pass `pass` inside its leaf for EVERY slot of the pass (then it needs no bound check)?
template <int M>
constexpr bool lp_step_always(int pass, int t) {
    for (int g = 0; g < 8; ++g) {
        const int s = pass * 8 + g;
        if (s >= LongPlanOf<M>::value.nslot) break;
        const int cA = lp_e1<M>(s) - lp_e0<M>(s), cB = lp_e2<M>(s) - lp_e1<M>(s);
        const bool in = lp_two<M>(s) ? (t < 8 ? t < cA : t - 8 < cB) : t < cA;
        if (!in) return false;
    }
    return true;
}
template <int M>
constexpr bool lp_pass_has_pair(int pass) {
    for (int g = 0; g < 8; ++g)
        if (pass * 8 + g < LongPlanOf<M>::value.nslot && lp_two<M>(pass * 8 + g)) return true;
    return false;
}

template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
    if constexpr (N > 0) {
        sfor<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int FUN, int M, int BATCH = 8>
__device__ __forceinline__ void row_reduce_long(const double *U, int l, double &sa, double &sb) {
    using O = Obj<FUN>;
    constexpr bool TWO = O::TWO, BMUL = O::BMUL;
    using PL = LongPlanOf<M>;
    constexpr int NLEAF = PL::value.nleaf, NSLOT = PL::value.nslot, NPASS = (NSLOT + 7) / 8;
    constexpr int TAIL = M % kGroup, T0 = (M / kGroup) * kGroup;
    const int j = l & (kGroup - 1), grp = l >> 3;
    const double identB = BMUL ? 1.0 : 0.0;
    double leafA[NLEAF], leafB[NLEAF];  // leaf sums, uniform over the wave (compile-time indices only: registers)
    // the tail terms (after the last leaf's tree), one per lane j < TAIL, requested before the passes
    double tailA = 0.0, tailB = identB;
    if constexpr (TAIL > 0) {
        const int e = T0 + (j < TAIL ? j : 0);
        O::term(U[e], O::NEXT ? U[e + 1] : 0.0, e, tailA, tailB);
    }
    sfor<NPASS>([&](auto pass_) {
        constexpr int PASS = decltype(pass_)::value;
        constexpr bool HASPAIR = lp_pass_has_pair<M>(PASS);
        // this lane's slot: first block of the first half (steps 0..7), "first block - 8" of the second half, the limits
        int a0 = 0, b0 = 0, lim1 = 16, lim2 = 16;
        bool two = false;
        sfor<8>([&](auto g_) {
            constexpr int G = decltype(g_)::value, S = PASS * 8 + G;
            if constexpr (S < NSLOT) {
                constexpr int E0 = lp_e0<M>(S), E1 = lp_e1<M>(S), E2 = lp_e2<M>(S);
                constexpr bool TW = lp_two<M>(S);
                if (grp == G) a0 = E0, b0 = TW ? E1 - 8 : E0, lim1 = E1 - E0, lim2 = TW ? 8 + (E2 - E1) : E1 - E0, two = TW;
            }
        });  // (groups beyond the last slot walk over slot 0's blocks; nothing of theirs is kept)
        const double *Ua = U + a0 * kGroup + j, *Ub = U + b0 * kGroup + j;
        double chA = 0.0, chB = identB, firstA = 0.0, firstB = identB;
        static_assert(BATCH == 8 || BATCH == 4, "steps whose LDS reads are in flight together");
        sfor<kLeafBlocks / BATCH>([&](auto q_) {
            constexpr int Q = decltype(q_)::value, H = Q * BATCH >= 8 ? 1 : 0;  // H: second half (the pair's second leaf)
            double x[BATCH], xn[BATCH];
#pragma unroll
            for (int t = 0; t < BATCH; ++t) {
                const double *q = (H ? Ub : Ua) + (BATCH * Q + t) * kGroup;
                x[t] = q[0];
                xn[t] = O::NEXT ? q[1] : 0.0;
            }
            if constexpr (Q * BATCH == 8 && HASPAIR) {  // the first leaf of a pair is complete: its tree; the chains start again
                firstA = group_tree<false>(chA);
                firstB = TWO ? group_tree<BMUL>(chB) : identB;
            }
            if constexpr (BATCH < 8) __builtin_amdgcn_sched_barrier(0);  // (keeps the batches' cosines from being interleaved: registers)
            sfor<BATCH>([&](auto t_) {
                constexpr int T = BATCH * Q + decltype(t_)::value;
                double a, b;
                O::term(x[T - BATCH * Q], xn[T - BATCH * Q], ((H ? b0 : a0) + T) * kGroup + j, a, b);
                if constexpr (T == 0) {
                    chA = a, chB = b;
                } else {
                    double nA = chA + a, nB = TWO ? combine<BMUL>(chB, b) : identB;
                    if constexpr (T == 8 && HASPAIR) nA = two ? a : nA, nB = two ? b : nB;
                    if constexpr (lp_step_always<M>(PASS, T)) {
                        chA = nA, chB = nB;
                    } else {
                        const bool in = T < (H ? lim2 : lim1);
                        chA = in ? nA : chA, chB = in ? nB : chB;
                    }
                }
            });
        });
        const double curA = group_tree<false>(chA);
        const double curB = TWO ? group_tree<BMUL>(chB) : identB;
        sfor<8>([&](auto g_) {
            constexpr int G = decltype(g_)::value, S = PASS * 8 + G;
            if constexpr (S < NSLOT) {
                constexpr int F = PL::value.sfirst[S];
                if constexpr (lp_two<M>(S)) {
                    leafA[F] = readlane_f64(firstA, 8 * G);
                    leafB[F] = TWO ? readlane_f64(firstB, 8 * G) : identB;
                    leafA[F + 1] = readlane_f64(curA, 8 * G);
                    leafB[F + 1] = TWO ? readlane_f64(curB, 8 * G) : identB;
                } else {
                    leafA[F] = readlane_f64(curA, 8 * G);
                    leafB[F] = TWO ? readlane_f64(curB, 8 * G) : identB;
                }
            }
        });
    });
    if constexpr (TAIL > 0) {
        sfor<TAIL>([&](auto k_) {
            constexpr int K = decltype(k_)::value;
            leafA[NLEAF - 1] = leafA[NLEAF - 1] + readlane_f64(tailA, K);
            if (TWO) leafB[NLEAF - 1] = combine<BMUL>(leafB[NLEAF - 1], readlane_f64(tailB, K));
        });
    }
    sfor<PL::value.nmerge>([&](auto m_) {
        constexpr int MM = decltype(m_)::value;
        constexpr int LEFT = PL::value.mleft[MM], RIGHT = PL::value.mright[MM];
        leafA[LEFT] = leafA[LEFT] + leafA[RIGHT];
        if (TWO) leafB[LEFT] = combine<BMUL>(leafB[LEFT], leafB[RIGHT]);
    });
    sa = 0.0 + leafA[0];
    sb = (TWO && !BMUL) ? 0.0 + leafB[0] : leafB[0];
}

// lexicographic (value, index) minimum = np.argmin's first-minimum rule
__device__ __forceinline__ void argmin_combine(double &f, int64_t &i, double f2, int64_t i2) {
    if (f2 < f || (f2 == f && i2 < i)) {
        f = f2;
        i = i2;
    }
}

constexpr int kDppRowMirror = 0x140;  // row_mirror: lane i <-> 15-i inside each 16 lanes

// minimum over the 64 lanes, returned to every lane: four DPP steps + four readlanes (no LDS traffic)
__device__ __forceinline__ double wave_min_f64(double v) {
    v = fmin(v, dpp_f64<kDppXor1>(v));
    v = fmin(v, dpp_f64<kDppXor2>(v));
    v = fmin(v, dpp_f64<kDppHalfMirror>(v));
    v = fmin(v, dpp_f64<kDppRowMirror>(v));  // every 16-lane row is uniform now
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// maximum over the 64 lanes, returned to every lane
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<kDppXor1>(v));
    v = fmax(v, dpp_f64<kDppXor2>(v));
    v = fmax(v, dpp_f64<kDppHalfMirror>(v));
    v = fmax(v, dpp_f64<kDppRowMirror>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// row_sum<64>'s butterfly (v += lane ^ 1, ^ 2, ^ 4, ^ 8, ^ 16, ^ 32) without LDS traffic.  After the step with lane ^ k both
// partners hold the same bits (a + b == b + a), so groups of 2k lanes are uniform: the mirrors stand in for ^ 4 and ^ 8, and
// the last two steps take the other rows' values through readlane.  Same operands in every addition: same bits.
__device__ __forceinline__ double wave_sum_butterfly(double v, int lane) {
    v = v + dpp_f64<kDppXor1>(v);
    v = v + dpp_f64<kDppXor2>(v);
    v = v + dpp_f64<kDppHalfMirror>(v);
    v = v + dpp_f64<kDppRowMirror>(v);
    const double s0 = readlane_f64(v, 0), s1 = readlane_f64(v, 16), s2 = readlane_f64(v, 32), s3 = readlane_f64(v, 48);
    const double partner = (lane & 16) ? ((lane & 32) ? s2 : s0) : ((lane & 32) ? s3 : s1);
    v = v + partner;
    const double lo = readlane_f64(v, 0), hi = readlane_f64(v, 32);
    return v + ((lane & 32) ? lo : hi);
}

// What the best / termination kernel of a CPSO graph can say about the swarm radius R = max_i ||X_i - g_new|| from the
// generation kernel's r = max_i ||X_i - g_old|| and the step of the best d = ||g_new - g_old|| (its dx):
//   d == 0 (the best did not move: g_new IS g_old, bit for bit)  ->  R = r exactly, same operations as pso_radius_kernel;
//   otherwise |R - r| <= d (triangle inequality), so r - d above / r + d below the threshold delta * sqrt(4n) by a relative
//   margin of 1e-6 (rounding is 1e-13) decides `radius < delta` without the radius; else the radius pass runs.
// rdec[0] = one of these, rdec[1] = bits of r.
constexpr unsigned long long kRadiusExactNeeded = 0ull, kRadiusKnown = 1ull, kRadiusAbove = 2ull, kRadiusBelow = 3ull;
__device__ __forceinline__ unsigned long long radius_decision(double r, double d, double delta, int n) {
    if (d == 0.0) return kRadiusKnown;
    const double thr = delta * sqrt(4.0 * (double)n);
    if (r - d > thr * (1.0 + 1.0e-6)) return kRadiusAbove;
    if (r + d < thr * (1.0 - 1.0e-6)) return kRadiusBelow;
    return kRadiusExactNeeded;
}

// (min f, its index) over the wave when LOWER LANES HOLD LOWER INDICES: the first lane that holds the
// minimum wins = np.argmin's first-minimum rule.  Result in every lane.
// One-batch DE rows (the metric shape): the 4 MB a generation writes are next read after the kernel boundary, and as
// streaming (nt) stores they do not sit dirty in the XCDs' L2s until the end-of-kernel write-back: 7.45 -> 6.95 us per
// generation (profiles/r4_de_m_store_flavours.txt: sc1 / sc0 sc1 write-through -1 %, uncached buffers +2.6 %, nt LOADS
// of the own / donor rows +1 % / +8 %).  NOT for long rows or PSO: n=1024 P=16384 90 -> 96 us, C3a / C3b unchanged.
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) {
    __builtin_nontemporal_store(v, p);
}

__device__ __forceinline__ void wave_argmin_ordered(double &f, int64_t &i) {
    const double m = wave_min_f64(f);
    const unsigned long long mask = __ballot(f == m);
    const int src = mask ? (int)__ffsll((long long)mask) - 1 : 0;  // all NaN: lane 0, as a sequential scan would
    const int lo = __builtin_amdgcn_readlane((int)(i & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(i >> 32), src);
    f = m;
    i = ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
}

// smallest of a 64-bit signed value over the wave, in every lane (DPP + readlanes, like wave_min_f64)
template <int CTRL>
__device__ __forceinline__ int64_t dpp_i64(int64_t v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(v & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
}
__device__ __forceinline__ int64_t min_i64(int64_t a, int64_t b) { return b < a ? b : a; }
__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
    v = min_i64(v, dpp_i64<kDppXor1>(v));
    v = min_i64(v, dpp_i64<kDppXor2>(v));
    v = min_i64(v, dpp_i64<kDppHalfMirror>(v));
    v = min_i64(v, dpp_i64<kDppRowMirror>(v));
    const unsigned long long u = (unsigned long long)v;
    const int64_t a = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 0) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)u, 0));
    const int64_t b = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 16) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)u, 16));
    const int64_t c = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 32) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)u, 32));
    const int64_t d = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 48) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)u, 48));
    return min_i64(min_i64(a, b), min_i64(c, d));
}

// (min f, smallest index that holds it) over the wave for ANY assignment of indices to lanes, in every lane: the
// minimum value, then the minimum index among the lanes that hold it -- two DPP reductions instead of six rounds of
// ds_bpermute exchanges.  (All NaN: index 0, as np.argmin.)
__device__ __forceinline__ void wave_argmin_all(double &f, int64_t &i) {
    const double m = wave_min_f64(f);
    const int64_t r = wave_min_i64(f == m ? i : INT64_MAX);
    f = m;
    i = r == INT64_MAX ? 0 : r;
}

}  // namespace sx
