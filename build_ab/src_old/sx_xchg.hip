// Peer exchange buffers for the multi-GPU generation kernels: allocation, IPC export / import,
// the transport self-test and the host-side record decode.
//
// Takes the place of the reference's per-generation MPI traffic (stochopy/optimize/_common.py:58-72):
// each rank's best-of-generation record is written straight into its peers' HBM over xGMI by the
// generation kernel itself (csrc/sx_de.hip, XM = 2); see sx_xchg.hpp for the wire format.
#include <cstring>
#include <vector>

#include "sx_host.hpp"
#include "sx_xchg.hpp"

using namespace sx;

namespace {

__device__ __forceinline__ uint32_t probe_word(uint32_t round, uint32_t src, uint32_t j) {
    uint32_t h = round * 0x9E3779B9u ^ (src + 1u) * 0x85EBCA6Bu ^ (j + 1u) * 0xC2B2AE35u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h;
}

// One workgroup: wave w writes this rank's pattern of round `tag` into peers w, w+nw, ...; then the whole
// workgroup waits for every rank's pattern in its own probe slots and checks every word.  Rounds alternate
// between the two halves of a slot (a rank starts round r+2 only after it has seen every peer's round r+1,
// which a peer sends only after it has finished checking round r: the generation kernels' own argument).
__global__ __launch_bounds__(512) void xchg_probe_kernel(const sx_xchg_args x, int n, uint32_t tag, int *result) {
    const int64_t sw = xchg_slot_words(n) / 2;
    const int64_t half = (int64_t)(tag & 1u) * sw;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63), nw = (int)(blockDim.x >> 6);
    for (int r = wave; r < x.world; r += nw) {
        uint64_t *dst = x.peer[r] + xchg_probe_offset(n, x.rank) + half;
        for (int64_t j = lane; j < sw; j += kWave) ll_store(dst + j, probe_word(tag, (uint32_t)x.rank, (uint32_t)j), tag);
    }
    const uint64_t *own = x.peer[x.rank];
    const uint64_t t0 = wall_clock64();
    int bad = 0;
    for (int64_t k = threadIdx.x; k < (int64_t)x.world * sw; k += blockDim.x) {
        const int src = (int)(k / sw);
        const int64_t j = k % sw;
        const uint64_t *p = own + xchg_probe_offset(n, src) + half + j;
        uint64_t w;
        for (;;) {
            w = ll_load(p);
            if (ll_ok(w, tag)) break;
            if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (!bad && (uint32_t)w != probe_word(tag, (uint32_t)src, (uint32_t)j)) bad = 1;
        if (bad) break;
    }
    if (bad) atomicExch(result, 1);
}

}  // namespace

extern "C" int64_t sx_xchg_bytes(int world, int n) {
    if (world < 1 || world > SX_MAX_PEERS || n < 1) return -1;
    return xchg_total_words(n) * (int64_t)sizeof(uint64_t);
}

extern "C" int64_t sx_xchg_relay_bytes(int n) {
    if (n < 1) return -1;
    return 2 * xchg_relay_stride(n) * (int64_t)sizeof(uint64_t);
}

extern "C" int sx_xchg_alloc(int64_t bytes, void **ptr, void *handle) {
    SX_REQUIRE(ptr != nullptr && handle != nullptr && bytes > 0, "sx_xchg_alloc: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == SX_IPC_HANDLE_BYTES, "IPC handle size");
    void *p = nullptr;
    // uncached device memory: peers' writes land in HBM and local readers never see a stale cache line
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        SX_HIP(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained));
    }
    SX_HIP(hipMemset(p, 0, (size_t)bytes));
    SX_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return hip_fail(e, "hipIpcGetMemHandle", __FILE__, __LINE__);
    }
    std::memcpy(handle, &h, sizeof h);
    *ptr = p;
    return 0;
}

extern "C" int sx_pop_alloc(int64_t bytes, void **ptr, void *handle) {
    SX_REQUIRE(ptr != nullptr && handle != nullptr && bytes > 0, "sx_pop_alloc: bad arguments");
    void *p = nullptr;
    SX_HIP(hipMalloc(&p, (size_t)bytes));
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return hip_fail(e, "hipIpcGetMemHandle", __FILE__, __LINE__);
    }
    std::memcpy(handle, &h, sizeof h);
    *ptr = p;
    return 0;
}

extern "C" int sx_xchg_free(void *ptr) {
    if (ptr) SX_HIP(hipFree(ptr));
    return 0;
}

extern "C" int sx_xchg_open(const void *handle, void **ptr) {
    SX_REQUIRE(handle != nullptr && ptr != nullptr, "sx_xchg_open: bad arguments");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void *p = nullptr;
    SX_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *ptr = p;
    return 0;
}

extern "C" int sx_xchg_close(void *ptr) {
    if (ptr) SX_HIP(hipIpcCloseMemHandle(ptr));
    return 0;
}

static int check_xchg(const sx_xchg_args *x, const char *who) {
    SX_REQUIRE(x != nullptr, "sx_xchg: null exchange arguments");
    SX_REQUIRE(x->world >= 1 && x->world <= SX_MAX_PEERS && x->rank >= 0 && x->rank < x->world,
               "sx_xchg: bad world / rank (at most 8 ranks)");
    SX_REQUIRE(x->error != nullptr && x->timeout_ticks > 0 && x->relay != nullptr,
               "sx_xchg: error word / timeout / relay buffer missing");
    for (int r = 0; r < x->world; ++r) SX_REQUIRE(x->peer[r] != nullptr, "sx_xchg: unmapped peer buffer");
    (void)who;
    return 0;
}
namespace sx {
int check_xchg_args(const sx_xchg_args *x) { return check_xchg(x, "sx_xchg"); }
}

extern "C" int sx_xchg_probe(const sx_xchg_args *x, int n, int rounds, void *stream) {
    if (int rc = check_xchg(x, "sx_xchg_probe")) return rc;
    SX_REQUIRE(n >= 1 && rounds >= 1, "sx_xchg_probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int *d_res = nullptr;
    SX_HIP(hipMalloc((void **)&d_res, sizeof(int)));
    SX_HIP(hipMemsetAsync(d_res, 0, sizeof(int), s));
    for (int r = 1; r <= rounds; ++r)
        hipLaunchKernelGGL(xchg_probe_kernel, dim3(1), dim3(512), 0, s, *x, n, (uint32_t)r, d_res);
    int res = 1;
    hipError_t e = hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_res);
    if (e != hipSuccess) return hip_fail(e, "sx_xchg_probe", __FILE__, __LINE__);
    return res;
}

extern "C" int sx_xchg_read_record(const sx_xchg_args *x, int n, int parity, int src, double *record, void *stream) {
    if (int rc = check_xchg(x, "sx_xchg_read_record")) return rc;
    SX_REQUIRE(record != nullptr && n >= 1 && (parity == 0 || parity == 1) && src >= 0 && src < x->world,
               "sx_xchg_read_record: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    std::vector<uint64_t> w((size_t)(2 * (n + 2)));
    // the kernels only wait for the words they use (rand* strategies never read the row), so the tail of a
    // record may still be in flight for a few microseconds: re-read until all words carry one tag
    for (int attempt = 0; attempt < 64; ++attempt) {
        SX_HIP(hipMemcpyAsync(w.data(), x->peer[x->rank] + xchg_slot_offset(n, parity, src),
                              w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        const uint32_t tag = (uint32_t)(w[0] >> 32);
        bool whole = true;
        for (size_t j = 0; j < w.size(); ++j) whole = whole && (uint32_t)(w[j] >> 32) == tag;
        if (!whole) continue;
        for (int j = 0; j < n + 2; ++j) {
            const uint64_t b = (w[2 * j + 1] << 32) | (w[2 * j] & 0xffffffffull);
            std::memcpy(record + j, &b, sizeof b);
        }
        return 0;
    }
    set_error("sx_xchg_read_record: record is torn (mixed generation tags)");
    return -1;
}

// ---------------------------------------------------------------------------
// Best-of-generation over all ranks as ONE one-workgroup kernel (for generation kernels that are not chained:
// PSO / CPSO, the DE two-kernel path): what sx_shard_best + an all-gather + sx_gather_finalize do over RCCL.
//   records of this shard -> shard best -> its record into every peer's slot (one wavefront per peer) ->
//   wait for all ranks' records of this generation -> global best (lowest f, ties to the lowest global row)
//   -> dx, gbest, status, it++ (_common.py:131-158), identically on every rank.
// Slots are double-buffered by generation parity; a rank can be at most one generation ahead of a peer (it
// needs every peer's record of generation g to produce generation g+1).
// ---------------------------------------------------------------------------
namespace {
constexpr int kXfThreads = 512;

__global__ __launch_bounds__(kXfThreads) void xchg_finalize_kernel(
    const double *__restrict__ part_f, const int64_t *__restrict__ part_i, int64_t npart,
    const double *__restrict__ rows0, const double *__restrict__ rows1, int64_t ld, int n, int64_t row0,
    double *__restrict__ gbest, sx_state *__restrict__ state, int maxiter, double xtol, double ftol,
    const sx_xchg_args x) {
    __shared__ double sf[kXfThreads / kWave];
    __shared__ int64_t si[kXfThreads / kWave];
    __shared__ double sd[kXfThreads / kWave];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = kXfThreads / kWave;
    // records first (they do not depend on the state word), then the state
    double bf = __builtin_huge_val();
    int64_t bi = INT64_MAX;
    for (int64_t k0 = tid; k0 < npart; k0 += (int64_t)kXfThreads * 8) {
        double f[8];
        int64_t i[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t k = k0 + (int64_t)u * kXfThreads;
            f[u] = k < npart ? part_f[k] : __builtin_huge_val();
            i[u] = k < npart ? part_i[k] : INT64_MAX;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (f[u] < bf || (f[u] == bf && i[u] < bi)) {
                bf = f[u];
                bi = i[u];
            }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double f2 = __shfl_xor(bf, off, kWave);
        const int64_t i2 = __shfl_xor((long long)bi, off, kWave);
        if (f2 < bf || (f2 == bf && i2 < bi)) {
            bf = f2;
            bi = i2;
        }
    }
    if (lane == 0) {
        sf[wave] = bf;
        si[wave] = bi;
    }
    __syncthreads();
    for (int w = 0; w < nw; ++w)
        if (sf[w] < bf || (sf[w] == bf && si[w] < bi)) {
            bf = sf[w];
            bi = si[w];
        }
    if (state->done || *x.error) return;  // uniform
    const int64_t it = state->it + 1;     // the generation being finalised
    const uint32_t tag = (uint32_t)(it + 1);
    const int parity = (int)(it & 1);
    const double *row = ((it & 1) ? rows1 : rows0) + bi * ld;
    for (int r = wave; r < x.world; r += nw)
        xchg_push_record(x.peer[r] + xchg_slot_offset(n, parity, x.rank), bf, row0 + bi, row, n, tag, lane);
    double gf;
    int64_t gi;
    int winner;
    if (!xchg_wait_best(x.peer[x.rank] + xchg_slot_offset(n, parity, 0), n, x.world, tag, x.timeout_ticks, lane, gf, gi,
                        winner)) {
        if (lane == 0) atomicExch(x.error, 1);
        return;
    }
    // the winner's row out of its slot (tagged words; a word still in flight is re-read), dx, gbest
    const uint64_t *src = x.peer[x.rank] + xchg_slot_offset(n, parity, winner) + 4;
    const uint64_t t0 = wall_clock64();
    double acc = 0.0;
    for (int e = tid; e < n; e += kXfThreads) {
        uint64_t lo, hi;
        for (;;) {
            lo = ll_load(src + 2 * e);
            hi = ll_load(src + 2 * e + 1);
            if (ll_ok(lo, tag) && ll_ok(hi, tag)) break;
            if ((int64_t)(wall_clock64() - t0) > x.timeout_ticks) {
                atomicExch(x.error, 1);
                break;
            }
        }
        const double v = ll_join_f64(lo, hi);
        const double d = gbest[e] - v;
        acc += d * d;
        gbest[e] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if (lane == 0) sd[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double ss = 0.0;
        for (int w = 0; w < nw; ++w) ss += sd[w];
        const double dx = sqrt(ss);
        int status = SX_STATUS_NONE;
        if (dx <= xtol && gf <= ftol)
            status = 0;
        else if (gf <= ftol)
            status = 1;
        else if (it >= maxiter)
            status = -1;
        state->it = it;
        state->gbidx = gi;
        state->gfit = gf;
        state->dx = dx;
        state->status = status;
        state->done = status != SX_STATUS_NONE;
    }
}
}  // namespace

extern "C" int sx_xchg_finalize(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                                const double *rows1, int64_t ld, int n, int64_t row0, double *gbest, sx_state *state,
                                int maxiter, double xtol, double ftol, const sx_xchg_args *x, void *stream) {
    if (int rc = check_xchg(x, "sx_xchg_finalize")) return rc;
    SX_REQUIRE(part_f && part_i && rows0 && rows1 && gbest && state && npart >= 1 && n >= 1 && row0 >= 0,
               "sx_xchg_finalize: bad arguments");
    hipLaunchKernelGGL(xchg_finalize_kernel, dim3(1), dim3(kXfThreads), 0, (hipStream_t)stream, part_f, part_i, npart,
                       rows0, rows1, ld, n, row0, gbest, state, maxiter, xtol, ftol, *x);
    SX_LAUNCH_CHECK();
    return 0;
}
