// Peer exchange over xGMI: tagged 8-byte words ("data + generation tag in one store").
//
// A rank's record [f, global row, best row[0..n)] travels as 2*(n+2) words; word 2j / 2j+1 carry the low /
// high half of double j together with the 32-bit tag of the generation.  An aligned 8-byte store is a
// single transaction, so a reader that sees the expected tag in a word also sees its data: arrival is
// detected on the data itself, with no separate flag and no release fence between data and flag (one
// one-way trip over the link instead of a round trip plus a trip).  The buffers are uncached device
// memory and every access is a system-scope atomic, so neither side's caches are involved.
#pragma once
#include "sx_device.hpp"

namespace sx {

// words per slot, padded to whole 128-byte lines
__host__ __device__ inline int64_t xchg_slot_words(int n) { return ((2 * ((int64_t)n + 2) + 15) / 16) * 16; }
// buffer = slots[2][SX_MAX_PEERS][slot_words] then probe[SX_MAX_PEERS][slot_words]
__host__ __device__ inline int64_t xchg_slot_offset(int n, int parity, int src) {
    return ((int64_t)parity * SX_MAX_PEERS + src) * xchg_slot_words(n);
}
__host__ __device__ inline int64_t xchg_probe_offset(int n, int src) {
    return ((int64_t)2 * SX_MAX_PEERS + src) * xchg_slot_words(n);
}
__host__ __device__ inline int64_t xchg_total_words(int n) { return (int64_t)3 * SX_MAX_PEERS * xchg_slot_words(n); }
// relay (ordinary memory): per generation parity one record as tagged words, then the "ready" word (its own line)
__host__ __device__ inline int64_t xchg_relay_stride(int n) { return xchg_slot_words(n) + 16; }

__device__ __forceinline__ void ll_store(uint64_t *p, uint32_t data, uint32_t tag) {
    __hip_atomic_store(p, ((uint64_t)tag << 32) | (uint64_t)data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t ll_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ll_store_f64(uint64_t *p2, double v, uint32_t tag) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    ll_store(p2, (uint32_t)b, tag);
    ll_store(p2 + 1, (uint32_t)(b >> 32), tag);
}
__device__ __forceinline__ double ll_join_f64(uint64_t lo, uint64_t hi) {
    return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
}
__device__ __forceinline__ bool ll_ok(uint64_t w, uint32_t tag) { return (uint32_t)(w >> 32) == tag; }

// One wavefront writes this rank's record for generation tag `tag` into dst (a peer's slot for this rank).
__device__ __forceinline__ void xchg_push_record(uint64_t *dst, double f, int64_t grow, const double *__restrict__ row,
                                                 int n, uint32_t tag, int lane) {
    // the header goes first: the readers learn the winner while the row is still being fetched here (they
    // re-read any row word that has not arrived yet)
    if (lane < 2) ll_store_f64(dst + 2 * lane, lane == 0 ? f : __longlong_as_double((long long)grow), tag);
    for (int j0 = lane; j0 < n; j0 += 8 * kWave) {  // 8 row loads in flight per trip, then their stores
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * kWave;
            v[u] = j < n ? row[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * kWave;
            if (j < n) ll_store_f64(dst + 2 * (j + 2), v[u], tag);
        }
    }
}

// Every lane of the wave: wait for the (f, global row) header words of all `world` records of generation
// `tag` in this rank's own slots, then pick the global best -- lowest f, ties to the lowest global row
// (= np.argmin over the whole population, _common.py:132; ranks own ascending row ranges).
// Returns false on timeout.  winner = rank whose record holds the best row.
__device__ __forceinline__ bool xchg_wait_best(const uint64_t *slots_p, int n, int world, uint32_t tag,
                                               int64_t timeout_ticks, int lane, double &bf, int64_t &bi, int &winner) {
    const int64_t sw = xchg_slot_words(n);
    const bool mine = lane < 4 * world;
    const uint64_t *p = slots_p + (int64_t)(lane >> 2) * sw + (lane & 3);
    const uint64_t t0 = wall_clock64();
    uint64_t w = 0;
    for (;;) {
        if (mine) w = ll_load(p);
        if (__all(!mine || ll_ok(w, tag))) break;
        if ((int64_t)(wall_clock64() - t0) > timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    const int data = (int)(uint32_t)w;
    const int r4 = (lane & 7) * 4;  // lane r < world assembles record r
    const uint32_t f_lo = (uint32_t)__shfl(data, r4 + 0, kWave), f_hi = (uint32_t)__shfl(data, r4 + 1, kWave);
    const uint32_t i_lo = (uint32_t)__shfl(data, r4 + 2, kWave), i_hi = (uint32_t)__shfl(data, r4 + 3, kWave);
    const bool rec = lane < world;
    double f = rec ? __longlong_as_double((long long)(((uint64_t)f_hi << 32) | f_lo)) : __builtin_huge_val();
    int64_t i = rec ? (int64_t)(((uint64_t)i_hi << 32) | i_lo) : INT64_MAX;
    const double m = wave_min_f64(f);
    const unsigned long long mask = __ballot(rec && f == m);
    const int src = mask ? (int)__ffsll((long long)mask) - 1 : 0;  // NaN everywhere: rank 0, like argmin's first
    const int lo = __builtin_amdgcn_readlane((int)(i & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(i >> 32), src);
    bf = readlane_f64(f, src);
    bi = ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
    winner = src;
    return true;
}

}  // namespace sx
