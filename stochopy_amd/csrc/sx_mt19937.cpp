// numpy-legacy random stream, host side (parity mode).
//
// The reference seeds numpy's *legacy global* generator (np.random.seed, at
// stochopy/optimize/de/_de.py:148-149, cpso/_cpso.py:153-154,
// cmaes/_cmaes.py:116-117) and draws through rand / uniform / randn / randint /
// permutation.  numpy is a third-party dependency of the reference (unpinned,
// setup.cfg:27-30; 2.2.6 in the build container); its legacy stream is the
// published MT19937 of Matsumoto & Nishimura plus five small derivations, which
// are restated here from their specification:
//   double     ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two consecutive words
//   uniform    lo + (hi - lo) * double
//   gauss      Marsaglia polar method, second variate cached across calls
//   bounded    mask = next_pow2(max) - 1; redraw (word & mask) until <= max
//   shuffle    Fisher-Yates from the end, j = bounded(i)
// Pinned bit-for-bit against numpy by tests/test_host_cpu.py and
// tests/golden/rng_stream.json.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#define SX_HAVE_X86 1
#define SX_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vpopcntdq,bmi2,popcnt,lzcnt,bmi")))
#else
#define SX_HAVE_X86 0
#endif

#include "../../include/stochopy_hip.h"

struct sx_mt {
    uint32_t mt[624];
    int pos;
    int has_gauss;
    double gauss;
    uint32_t out[624];  // the tempered words of the current block (filled by mt_twist; not part of the state)
    int out_valid;      // out[] matches mt[] (cleared whenever mt[] is set from outside)
};

namespace {

inline void mt_seed(sx_mt *g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->pos = 624;
    g->has_gauss = 0;
    g->gauss = 0.0;
    g->out_valid = 0;
}

inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

inline void mt_temper_block(sx_mt *g) {  // one vectorisable pass instead of 624 dependent call sites
    for (int k = 0; k < 624; ++k) g->out[k] = temper(g->mt[k]);
    g->out_valid = 1;
}

#if SX_HAVE_X86
// The wide forms below produce the SAME words (they are the same recurrence, 16 lanes at a time); SX_MT_SCALAR=1 keeps the
// scalar forms (tests compare the two).
inline bool have_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq") &&
                           __builtin_cpu_supports("avx512vpopcntdq") && __builtin_cpu_supports("bmi2") &&
                           std::getenv("SX_MT_SCALAR") == nullptr;
    return ok;
}

SX_AVX512 inline __m512i twist16(const uint32_t *mt, int k, int src) {
    const __m512i UP = _mm512_set1_epi32((int)0x80000000u), A = _mm512_set1_epi32((int)0x9908b0dfu),
                  one = _mm512_set1_epi32(1);
    const __m512i a = _mm512_loadu_si512(mt + k), b = _mm512_loadu_si512(mt + k + 1), c = _mm512_loadu_si512(mt + src);
    const __m512i y = _mm512_ternarylogic_epi32(UP, a, b, 0xCA);  // (a & UP) | (b & ~UP)
    const __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
    return _mm512_mask_xor_epi32(r, _mm512_test_epi32_mask(y, one), r, A);
}

SX_AVX512 void mt_twist_avx512(sx_mt *g) {
    uint32_t *mt = g->mt;
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    int k = 0;
    // words 0..226 read words k+1 (old) and k+397 (old): 14 whole vectors, then 3 words
    for (; k + 16 <= 624 - 397; k += 16) _mm512_storeu_si512(mt + k, twist16(mt, k, k + 397));
    for (; k < 624 - 397; ++k) {
        const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
        mt[k] = mt[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    // words 227..622 read k+1 (old: k+16 <= 623) and k-227 (new: 227 words behind, further than a vector)
    for (; k + 16 <= 623; k += 16) _mm512_storeu_si512(mt + k, twist16(mt, k, k - 227));
    for (; k < 623; ++k) {
        const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
        mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
    mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    g->pos = 0;
    const __m512i B = _mm512_set1_epi32((int)0x9d2c5680u), Cc = _mm512_set1_epi32((int)0xefc60000u);
    for (int q = 0; q < 624; q += 16) {  // 39 vectors exactly
        __m512i v = _mm512_loadu_si512(mt + q);
        v = _mm512_xor_si512(v, _mm512_srli_epi32(v, 11));
        v = _mm512_ternarylogic_epi32(v, _mm512_slli_epi32(v, 7), B, 0x78);   // v ^ (t & B)
        v = _mm512_ternarylogic_epi32(v, _mm512_slli_epi32(v, 15), Cc, 0x78);
        v = _mm512_xor_si512(v, _mm512_srli_epi32(v, 18));
        _mm512_storeu_si512(g->out + q, v);
    }
    g->out_valid = 1;
}
#endif

inline void mt_twist(sx_mt *g) {
#if SX_HAVE_X86
    if (have_avx512()) {
        mt_twist_avx512(g);
        return;
    }
#endif
    uint32_t *mt = g->mt;
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    int k = 0;
    for (; k < 624 - 397; ++k) {
        const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
        mt[k] = mt[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    for (; k < 623; ++k) {
        const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
        mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    }
    const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
    mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    g->pos = 0;
    mt_temper_block(g);
}

inline uint32_t next32(sx_mt *g) {
    if (g->pos == 624) mt_twist(g);
    else if (!g->out_valid) mt_temper_block(g);
    return g->out[g->pos++];
}

inline double next_double(sx_mt *g) {
    const uint32_t a = next32(g) >> 5, b = next32(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

inline uint64_t mask_for(uint64_t mx) {
    uint64_t m = mx;
    m |= m >> 1;
    m |= m >> 2;
    m |= m >> 4;
    m |= m >> 8;
    m |= m >> 16;
    m |= m >> 32;
    return m;
}

// value in [0, mx]; mx == 0 consumes nothing
inline uint64_t bounded(sx_mt *g, uint64_t mx, uint64_t mask) {
    if (mx == 0) return 0;
    uint64_t v;
    if (mx <= 0xffffffffull) {
        if (mx == 0xffffffffull) return next32(g);
        do {
            v = next32(g) & mask;
        } while (v > mx);
    } else {
        do {
            const uint64_t hi = next32(g), lo = next32(g);
            v = ((hi << 32) | lo) & mask;
        } while (v > mx);
    }
    return v;
}

// Fisher-Yates from the end for n <= 2^31 elements (the DE donor permutations, 16.8 million steps per
// generation at P = 4096): same words, same rejections as shuffle(), with the mask held per power-of-two range
// of i and 32-bit arithmetic throughout.
template <class T>
inline void shuffle_small(sx_mt *g, T *a, int32_t n) {
    int32_t i = n - 1;
    while (i >= 1) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        const int32_t lo = (int32_t)(mask >> 1) + 1;  // smallest i with this mask
        // one word per trip, no data-dependent branch: a rejected word (v > i) swaps a[i] with itself and
        // leaves i where it is (the rejection branch of the textbook loop mispredicts ~30 % of the time)
        while (i >= lo) {
            const uint32_t v = next32(g) & mask;
            const int32_t take = v <= (uint32_t)i;
            const int32_t j = take ? (int32_t)v : i;
            const T t = a[i];
            a[i] = a[j];
            a[j] = t;
            i -= take;
        }
    }
}

template <class T>
inline void shuffle(sx_mt *g, T *a, int64_t n) {
    if (n <= 0x7fffffff) {
        shuffle_small(g, a, (int32_t)n);
        return;
    }
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint64_t j = bounded(g, (uint64_t)i, mask_for((uint64_t)i));
        const T t = a[i];
        a[i] = a[j];
        a[j] = t;
    }
}

inline double next_gauss(sx_mt *g) {
    if (g->has_gauss) {
        const double t = g->gauss;
        g->has_gauss = 0;
        g->gauss = 0.0;
        return t;
    }
    double x1, x2, r2;
    do {
        x1 = 2.0 * next_double(g) - 1.0;
        x2 = 2.0 * next_double(g) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
    g->gauss = f * x1;
    g->has_gauss = 1;
    return f * x2;
}

}  // namespace

extern "C" sx_mt *sx_mt_create(uint32_t seed) {
    sx_mt *g = new sx_mt();
    mt_seed(g, seed);
    return g;
}
extern "C" void sx_mt_destroy(sx_mt *g) { delete g; }
extern "C" void sx_mt_seed(sx_mt *g, uint32_t seed) { mt_seed(g, seed); }

// count doubles, two consecutive words each: whole pairs of the current tempered block are converted in one
// vectorisable loop; a pair that straddles a block boundary goes through next_double()
static void fill_doubles(sx_mt *g, double *out, int64_t count) {
    int64_t i = 0;
    while (i < count) {
        if (g->pos >= 624 || !g->out_valid || 624 - g->pos < 2) {
            out[i++] = next_double(g);
            continue;
        }
        const int64_t pairs = (624 - g->pos) / 2;
        const int64_t take = pairs < count - i ? pairs : count - i;
        const uint32_t *w = g->out + g->pos;
        for (int64_t k = 0; k < take; ++k)
            out[i + k] = ((double)(w[2 * k] >> 5) * 67108864.0 + (double)(w[2 * k + 1] >> 6)) / 9007199254740992.0;
        g->pos += (int)(2 * take);
        i += take;
    }
}

extern "C" void sx_mt_random(sx_mt *g, double *out, int64_t count) { fill_doubles(g, out, count); }

extern "C" void sx_mt_uniform(sx_mt *g, double lo, double hi, double *out, int64_t count) {
    const double range = hi - lo;
    for (int64_t i = 0; i < count; ++i) out[i] = lo + range * next_double(g);
}

extern "C" void sx_mt_uniform_rows(sx_mt *g, const double *lo, const double *hi, int n, int64_t rows, double *out) {
    std::vector<double> range((size_t)n);
    for (int c = 0; c < n; ++c) range[c] = hi[c] - lo[c];
    for (int64_t r = 0; r < rows; ++r)
        for (int c = 0; c < n; ++c) out[r * n + c] = lo[c] + range[c] * next_double(g);
}

extern "C" void sx_mt_randn(sx_mt *g, double *out, int64_t count) {
    for (int64_t i = 0; i < count; ++i) out[i] = next_gauss(g);
}

extern "C" void sx_mt_randint(sx_mt *g, int64_t high, int64_t *out, int64_t count) {
    const uint64_t mx = (uint64_t)(high - 1);
    const uint64_t mask = mask_for(mx);
    for (int64_t i = 0; i < count; ++i) out[i] = (int64_t)bounded(g, mx, mask);
}

extern "C" void sx_mt_permutation(sx_mt *g, int64_t n, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    shuffle(g, out, n);
}

// _common.py:109-120 (lhs): x = rand(P, n) / P + linspace(-1, 1, P, endpoint=False)[:, None]; column j of the
// population is column j of x taken in the order of its own permutation(P); then * scale[j] + shift[j].  The caller
// passes numpy's own linspace / scale / shift vectors, so every number is formed by the operations numpy would use.
extern "C" void sx_mt_latin_hypercube(sx_mt *g, int64_t P, int n, const double *lin, const double *scale,
                                      const double *shift, double *out) {
    std::vector<double> x((size_t)(P * n));
    fill_doubles(g, x.data(), P * n);
    const double dP = (double)P;
    for (int64_t i = 0; i < P; ++i)
        for (int c = 0; c < n; ++c) x[i * n + c] = x[i * n + c] / dP + lin[i];
    std::vector<int64_t> perm((size_t)P);
    for (int c = 0; c < n; ++c) {
        for (int64_t i = 0; i < P; ++i) perm[i] = i;
        shuffle(g, perm.data(), P);
        for (int64_t i = 0; i < P; ++i) out[i * n + c] = x[perm[i] * n + c] * scale[c] + shift[c];
    }
}

#if SX_HAVE_X86
// The donor permutations without the permutations (round 4).  The reference shuffles arange(P-1) for every individual and
// keeps entries 0..k-1 (de/_de.py:304-311): 16.8 million Fisher-Yates steps per generation at P = 4096, one dependent
// load-swap-store each.  Two observations: (1) the SWAP TARGETS j_i (i = m-1 .. 1) depend only on the word stream -- masked
// rejection, a compare per word, 64 words at a time: word L is accepted iff v_L <= i - (#accepted words before L), which
// only the few words within 63 of the bound have to be asked one by one; (2) what ends up
// at position t is found by walking the swaps BACKWARDS from the last one (i = 1) to the first (i = m-1): q = t; at step i:
// q == i -> q = j_i, else q == j_i -> q = i; the array starts as the identity, so the entry is the final q.  q changes
// ~ln(m) times, so the walk is a vector search for the next step that touches it.  Same words consumed, same donors
// (tests/test_host_cpu.py compares with the scalar replay), no array of P-1 entries is ever shuffled.
// JT: the swap targets as 16-bit entries while they fit (P <= 65535: the walk then looks at 32 steps per compare), else 32-bit
template <class JT>
SX_AVX512 inline void store_targets(JT *dst, __m512i packed) {
    if constexpr (sizeof(JT) == 2)
        _mm256_storeu_si256((__m256i *)dst, _mm512_cvtepi32_epi16(packed));
    else
        _mm512_storeu_si512(dst, packed);
}

// steps of this block of the walk that touch position qq: J == qq (qq is swapped away) or i == qq (qq is the step's own index);
// lane b holds step base + b, whose index is i0 - b
template <class JT>
SX_AVX512 inline uint64_t walk_hits(__m512i Jv, int i0, uint32_t qq, uint64_t lanes, __m512i iota, __m512i iota16) {
    if constexpr (sizeof(JT) == 2) {
        const __m512i qv = _mm512_set1_epi16((short)qq);
        const __m512i Iv = _mm512_sub_epi16(_mm512_set1_epi16((short)i0), iota16);
        return ((uint64_t)_mm512_cmpeq_epi16_mask(Jv, qv) | (uint64_t)_mm512_cmpeq_epi16_mask(Iv, qv)) & lanes;
    } else {
        const __m512i qv = _mm512_set1_epi32((int)qq);
        const __m512i Iv = _mm512_sub_epi32(_mm512_set1_epi32(i0), iota);
        return ((uint64_t)_mm512_cmpeq_epi32_mask(Jv, qv) | (uint64_t)_mm512_cmpeq_epi32_mask(Iv, qv)) & lanes;
    }
}

template <class JT>
SX_AVX512 void de_donors_avx512(sx_mt *g, int64_t P, int k, int32_t *donors) {
    const int m = (int)(P - 1);
    constexpr int NL = 64 / (int)sizeof(JT);  // steps per compare in the walk
    std::vector<JT> store((size_t)m + 128, (JT)~(JT)0);
    JT *J = store.data() + 32;  // J[s] = j of step s (i = m-1-s); padding (all ones: no valid index) in front for the walk
    const __m512i iota = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m512i iota16 = _mm512_set_epi16(31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8,
                                            7, 6, 5, 4, 3, 2, 1, 0);
    for (int64_t ind = 0; ind < P; ++ind) {
        // ---- (1) the swap targets
        int s = 0, i = m - 1;
        while (i >= 1) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            const int lo = (int)(mask >> 1) + 1;  // smallest i with this mask
            if (i < 256) {
                while (i >= lo) {  // no data-dependent branch: a rejected word is overwritten by the next one
                    const uint32_t v = next32(g) & mask;
                    const int take = v <= (uint32_t)i;
                    J[s] = (JT)v;
                    s += take;
                    i -= take;
                }
                continue;
            }
            const __m512i vmask = _mm512_set1_epi32((int)mask);
            // 64 words per trip.  With i0 = i at the start of the trip, word L (a = accepted words before it, a <= L <= 63)
            // is accepted iff v_L <= i0 - a: certainly if v_L <= i0 - 63, certainly not if v_L > i0, whatever happened
            // before it -- eight compares that do not wait for anything but i0.  The few words in between (64 * 63 / mask of
            // a trip's words on average) are settled one by one from the accept mask so far.  (i >= 256 here; below that
            // most words are such words and the plain loop above is as good.)
            const int W = 64;
            while (i >= lo) {
                if (g->pos == 624) mt_twist(g);
                else if (!g->out_valid) mt_temper_block(g);
                const int avail = 624 - g->pos, cnt = avail < W ? avail : W;
                const uint32_t *w = g->out + g->pos;
                const __m512i hi_t = _mm512_set1_epi32(i), lo_t = _mm512_set1_epi32(i - (W - 1));
                __m512i v[4];
                uint64_t opt = 0, def = 0;
                const int nv = (cnt + 15) >> 4;
                if (cnt == 64) {  // the common trip: no lane masks, straight-line
                    v[0] = _mm512_and_si512(_mm512_loadu_si512(w), vmask);
                    v[1] = _mm512_and_si512(_mm512_loadu_si512(w + 16), vmask);
                    v[2] = _mm512_and_si512(_mm512_loadu_si512(w + 32), vmask);
                    v[3] = _mm512_and_si512(_mm512_loadu_si512(w + 48), vmask);
                    opt = (uint64_t)_mm512_cmple_epu32_mask(v[0], hi_t) | ((uint64_t)_mm512_cmple_epu32_mask(v[1], hi_t) << 16) |
                          ((uint64_t)_mm512_cmple_epu32_mask(v[2], hi_t) << 32) | ((uint64_t)_mm512_cmple_epu32_mask(v[3], hi_t) << 48);
                    // (word L of vector u has at most 16 u + 15 accepted words before it: a tighter bound per vector, fewer
                    //  words left to settle one by one)
                    def = (uint64_t)_mm512_cmple_epu32_mask(v[0], _mm512_set1_epi32(i - 15)) |
                          ((uint64_t)_mm512_cmple_epu32_mask(v[1], _mm512_set1_epi32(i - 31)) << 16) |
                          ((uint64_t)_mm512_cmple_epu32_mask(v[2], _mm512_set1_epi32(i - 47)) << 32) |
                          ((uint64_t)_mm512_cmple_epu32_mask(v[3], lo_t) << 48);
                } else {
                    for (int u = 0; u < nv; ++u) {
                        const int left = cnt - 16 * u;
                        const __mmask16 lanes = left >= 16 ? (__mmask16)0xffff : (__mmask16)((1u << left) - 1u);
                        v[u] = _mm512_and_si512(_mm512_maskz_loadu_epi32(lanes, w + 16 * u), vmask);
                        opt |= (uint64_t)_mm512_mask_cmple_epu32_mask(lanes, v[u], hi_t) << (16 * u);
                        def |= (uint64_t)_mm512_mask_cmple_epu32_mask(lanes, v[u], lo_t) << (16 * u);
                    }
                }
                uint64_t acc = def;
                for (uint64_t amb = opt & ~def; amb; amb &= amb - 1) {  // (no branch on the answer: it is a coin toss)
                    const int L = __builtin_ctzll(amb);
                    const int before = __builtin_popcountll(acc & ((1ull << L) - 1ull));
                    acc |= (uint64_t)((w[L] & mask) <= (uint32_t)(i - before)) << L;
                }
                const int na = __builtin_popcountll(acc);
                int used = cnt;
                if (na >= i - lo + 1) {
                    // the run (this mask) ends inside these words, at its (i - lo + 1)-th accepted one: what lies behind
                    // that word belongs to the next mask and is looked at again
                    const uint64_t last = _pdep_u64(1ull << (i - lo), acc);  // the (i - lo + 1)-th set bit of acc
                    used = __builtin_ctzll(last) + 1;
                    acc &= last | (last - 1);
                }
                const int taken = __builtin_popcountll(acc);
                // (compress to a register, then a plain store: the memory form of vpcompressd is slow on Zen)
                if (cnt == 64) {
                    const __mmask16 m0 = (__mmask16)acc, m1 = (__mmask16)(acc >> 16), m2 = (__mmask16)(acc >> 32),
                                    m3 = (__mmask16)(acc >> 48);
                    const int s1 = s + __builtin_popcount((unsigned)m0), s2 = s1 + __builtin_popcount((unsigned)m1),
                              s3 = s2 + __builtin_popcount((unsigned)m2);
                    store_targets(J + s, _mm512_maskz_compress_epi32(m0, v[0]));
                    store_targets(J + s1, _mm512_maskz_compress_epi32(m1, v[1]));
                    store_targets(J + s2, _mm512_maskz_compress_epi32(m2, v[2]));
                    store_targets(J + s3, _mm512_maskz_compress_epi32(m3, v[3]));
                    s += taken;
                } else {
                    for (int u = 0; u < nv; ++u) {
                        const __mmask16 m16 = (__mmask16)(acc >> (16 * u));
                        store_targets(J + s, _mm512_maskz_compress_epi32(m16, v[u]));
                        s += __builtin_popcount((unsigned)m16);
                    }
                }
                i -= taken;
                g->pos += used;
            }
        }
        // ---- (2) walk the swaps backwards (s = m-2 .. 0, i.e. i = 1 .. m-1) for positions 0..k-1
        uint32_t q[8];
        for (int t = 0; t < k; ++t) q[t] = (uint32_t)t;
        for (int hi = m - 2; hi >= 0; hi -= NL) {
            const int base = hi - (NL - 1);  // lanes b = 0..NL-1 <-> s = base + b (padding in front: all ones, lanes masked)
            const uint64_t all = NL == 32 ? 0xffffffffull : 0xffffull;
            const uint64_t lanes = base >= 0 ? all : (all << (-base)) & all;
            const __m512i Jv = _mm512_loadu_si512(J + base);
            for (int t = 0; t < k; ++t) {
                uint64_t h = walk_hits<JT>(Jv, m - 1 - base, q[t], lanes, iota, iota16);
                while (h) {
                    const int b = 63 - __builtin_clzll(h);  // the earliest of these steps in walking order: the largest s
                    const int ss = base + b;
                    const uint32_t ii = (uint32_t)(m - 1 - ss);
                    q[t] = q[t] == ii ? (uint32_t)J[ss] : ii;
                    h = walk_hits<JT>(Jv, m - 1 - base, q[t], lanes, iota, iota16) & ((1ull << b) - 1ull);
                }
            }
        }
        for (int t = 0; t < k; ++t) {
            const int32_t v = (int32_t)q[t];
            donors[(int64_t)t * P + ind] = v + (v >= ind ? 1 : 0);
        }
    }
}
#endif

extern "C" void sx_mt_de_donors(sx_mt *g, int64_t P, int k, int32_t *donors) {
    // individual i: permutation of arange(P) without i; entry t becomes donor t (de/_de.py:304-311)
#if SX_HAVE_X86
    if (have_avx512() && P >= 256 && P <= 0x40000000 && k <= 8) {
        // (SX_MT_WIDE_INDEX=1: the 32-bit index form for every P -- tests pin it against the scalar replay at small P, where
        //  it would otherwise only run for populations of more than 65535)
        if (P <= 65535 && std::getenv("SX_MT_WIDE_INDEX") == nullptr)
            de_donors_avx512<uint16_t>(g, P, k, donors);
        else
            de_donors_avx512<uint32_t>(g, P, k, donors);
        return;
    }
#endif
    std::vector<int32_t> a((size_t)(P - 1));
    for (int64_t i = 0; i < P; ++i) {
        for (int64_t v = 0; v < P - 1; ++v) a[v] = (int32_t)v;
        shuffle_small(g, a.data(), (int32_t)(P - 1));
        for (int t = 0; t < k; ++t) {
            const int32_t v = a[t];
            donors[(int64_t)t * P + i] = v + (v >= i ? 1 : 0);
        }
    }
}

extern "C" void sx_mt_de_async_draws(sx_mt *g, int64_t P, int k, int n, int32_t *donors, int32_t *irand,
                                     const double *lower, const double *upper, double *resample) {
    // de_async, per individual (de/_de.py:376-382): donor permutation, forced crossover index, Random's block
    std::vector<int32_t> a((size_t)(P - 1));
    const uint64_t mx = (uint64_t)(n - 1);
    const uint64_t mask = mask_for(mx);
    for (int64_t i = 0; i < P; ++i) {
        for (int64_t v = 0; v < P - 1; ++v) a[v] = (int32_t)v;
        shuffle_small(g, a.data(), (int32_t)(P - 1));
        for (int t = 0; t < k; ++t) {
            const int32_t v = a[t];
            donors[(int64_t)t * P + i] = v + (v >= i ? 1 : 0);
        }
        irand[i] = (int32_t)bounded(g, mx, mask);
        if (resample != nullptr)
            for (int c = 0; c < n; ++c) resample[i * n + c] = lower[c] + (upper[c] - lower[c]) * next_double(g);
    }
}

// np.random.get_state() / set_state() interchange: key[624], pos, has_gauss, cached_gaussian
extern "C" void sx_mt_get_state(sx_mt *g, uint32_t *key, int *pos, int *has_gauss, double *gauss) {
    std::memcpy(key, g->mt, sizeof g->mt);
    *pos = g->pos;
    *has_gauss = g->has_gauss;
    *gauss = g->gauss;
}
extern "C" void sx_mt_set_state(sx_mt *g, const uint32_t *key, int pos, int has_gauss, double gauss) {
    std::memcpy(g->mt, key, sizeof g->mt);
    g->out_valid = 0;
    g->pos = pos;
    g->has_gauss = has_gauss;
    g->gauss = gauss;
}
