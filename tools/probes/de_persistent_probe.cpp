// Round 4 probe: the metric shape (DE best1bin, n = 128, P = 4096: 256 workgroups of 16 rows, all
// co-resident) as ONE persistent kernel that hands generation g -> g+1 over INSIDE the launch with data-tagged workgroup
// records (one 8-byte {value, tag} granule per workgroup and generation, double-buffered by parity -- the protocol of
// csrc/sx_xchg.hpp at device scope), against the SAME skeleton with one kernel launch per generation (replayed hipGraph).
//
// The skeleton does a generation's memory work and dependencies, nothing else: every workgroup learns the best record of
// the previous generation (256 records -> min), then every row reads itself, two donor rows (hashed from row and
// generation) and the best row, writes the new row, and the workgroup publishes its record.  Every row depends on rows
// written by OTHER workgroups one generation earlier, so stale reads change the final checksum: all variants must print
// the same checksum as the per-launch form.
//
// Variants of the in-launch hand-off (MI355X_MICROARCH.md, "Valid forms"):
//   F  plain row stores -> __syncthreads -> lane-0 release fence (agent) -> s_waitcnt vmcnt(0) -> record;
//      consumer: wave 0 polls the 256 records (sc1 loads) -> lane-0 acquire fence (agent) -> __syncthreads -> plain loads
//   W  write-through row stores (sc1) -> s_waitcnt vmcnt(0) -> __syncthreads -> record; consumer as F
//   S  system-scope (sc0 sc1) row stores AND row loads, no fences at all (records polled with sc1 loads)
// Every poll has a time-out (the error word is reported): a lost workgroup cannot hang the GPU.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/de_persistent_probe.cpp -o build_ab/de_persistent_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 128, P = 4096, LPR = 32, WG = 256, ROWS = P / WG, THREADS = ROWS / 2 * 64;  // 16 rows, 8 waves
enum { LAUNCH = 0, FENCE = 1, WTHRU = 2, SYSTEM = 3 };

__device__ __forceinline__ uint64_t ld_sc1(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int V>
__device__ __forceinline__ double ld_row(const double *p) {
    if (V == SYSTEM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}
template <int V>
__device__ __forceinline__ void st_row(double *p, double v) {
    if (V == SYSTEM)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (V == WTHRU)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

struct Stamps {  // shader-clock stamps of workgroup 0 / thread 0 in the stamped generation (persistent variants)
    unsigned long long t[8];
};

// one generation of the skeleton for this workgroup.  rec[parity][WG]: {float bits of the workgroup's best value, tag = g}
template <int V>
__device__ __forceinline__ bool generation(const double *__restrict__ src, double *__restrict__ dst, uint64_t *rec, int g,
                                           int *err, long long timeout, Stamps *st) {
    __shared__ int s_best;
    __shared__ double s_val[ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & (LPR - 1), sub = lane / LPR;
    const int wg = blockIdx.x;
    const bool stamp = st != nullptr && wg == 0 && tid == 0;
    if (stamp) st->t[0] = clock64();
    // ---- records of generation g-1 (tag g): wave 0, four granules per lane, until all 256 carry the tag
    const uint64_t *rin = rec + (size_t)(g & 1) * WG;
    if (wave == 0) {
        uint64_t w[4];
        const long long t0 = wall_clock64();
        bool ok;
        do {
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = V == LAUNCH ? rin[lane * 4 + u] : ld_sc1(rin + lane * 4 + u);
            ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) ok = ok && (uint32_t)(w[u] >> 32) == (uint32_t)g;
            ok = __all(ok);
            if (!ok && wall_clock64() - t0 > timeout) {
                if (lane == 0) atomicExch(err, g + 1);
                break;
            }
        } while (!ok);
        // best = smallest (value bits, workgroup): the values are non-negative floats, so their bits order like they do
        uint64_t best = ~0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t key = ((w[u] & 0xffffffffull) << 32) | (uint64_t)(lane * 4 + u);
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) s_best = ok ? (int)(best & 0xffffffffull) * ROWS : -1;
        if (V == FENCE || V == WTHRU) {
            if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (stamp) st->t[1] = clock64();
    __syncthreads();
    const int brow = s_best;
    if (brow < 0) return false;
    if (stamp) st->t[2] = clock64();
    // ---- rows: own, two donors, best -> new row
    const int r = wave * 2 + sub, i = wg * ROWS + r;
    const unsigned h = (unsigned)i * 2654435761u + (unsigned)g * 40503u;
    const int d0 = (h >> 4) % P, d1 = (h >> 16) % P;
    double x[4], s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * LPR + l;
        const double xi = ld_row<V>(src + (size_t)i * N + e), a = ld_row<V>(src + (size_t)d0 * N + e),
                     b = ld_row<V>(src + (size_t)d1 * N + e), gb = ld_row<V>(src + (size_t)brow * N + e);
        const double u = ((h >> q) & 1) ? 0.25 * (gb + xi) + 0.5 * (a - b) : xi;
        x[q] = u;
        s += u * u;
    }
    if (stamp) st->t[3] = clock64();
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) st_row<V>(dst + (size_t)i * N + q * LPR + l, x[q]);
    if (l == 0) s_val[r] = s;
    if (V == WTHRU || V == SYSTEM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows have left
    if (stamp) st->t[4] = clock64();
    __syncthreads();
    // ---- the workgroup's record for generation g (tag g+1), written into the other parity
    if (tid == 0) {
        float m = (float)s_val[0];
        for (int k = 1; k < ROWS; ++k) m = fminf(m, (float)s_val[k]);
        if (V == FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const uint64_t g64 = ((uint64_t)(uint32_t)(g + 1) << 32) | (uint64_t)__float_as_uint(m);
        uint64_t *out = rec + (size_t)((g + 1) & 1) * WG + wg;
        if (V == LAUNCH)
            *out = g64;
        else
            __hip_atomic_store(out, g64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp) st->t[5] = clock64();
    return true;
}

template <int V>
__global__ __launch_bounds__(THREADS) void persistent(double *A, double *B, uint64_t *rec, int g0, int gens, int *err,
                                                      long long timeout, Stamps *st, int stamp_gen) {
    for (int g = g0; g < g0 + gens; ++g) {
        const double *src = (g & 1) ? B : A;
        double *dst = (g & 1) ? A : B;
        if (!generation<V>(src, dst, rec, g, err, timeout, g == stamp_gen ? st : nullptr)) return;
    }
}

__global__ __launch_bounds__(THREADS) void one_generation(double *A, double *B, uint64_t *rec, const int *gen_p, int k,
                                                          int *err) {
    const int g = *gen_p + k;  // (the generation number is a load after the boundary, as in the product kernel)
    const double *src = (g & 1) ? B : A;
    double *dst = (g & 1) ? A : B;
    (void)generation<LAUNCH>(src, dst, rec, g, err, 1ll << 40, nullptr);
}

// the per-launch skeleton with the two changes the product kernel could still take: OWN = the row's own elements requested
// before the generation number has arrived (the buffer parity is known from the launch's position in the graph), NT =
// streaming row stores
template <bool OWN, bool NT>
__global__ __launch_bounds__(THREADS) void one_generation_v(double *A, double *B, uint64_t *rec, const int *gen_p, int k,
                                                            int *err) {
    __shared__ int s_best;
    __shared__ double s_val[ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & (LPR - 1), sub = lane / LPR;
    const int wg = blockIdx.x, r = wave * 2 + sub, i = wg * ROWS + r;
    const int par = k & 1;  // graphs have an even number of generations and start at an even generation
    const double *src = par ? B : A;
    double *dst = par ? A : B;
    double xi[4];
    if (OWN) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xi[q] = src[(size_t)i * N + q * LPR + l];
    }
    const uint64_t *rin = rec + (size_t)par * WG;
    uint64_t w[4] = {0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = rin[lane * 4 + u];
    }
    const int g = *gen_p + k;
    if (wave == 0) {
        uint64_t best = ~0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t key = ((w[u] & 0xffffffffull) << 32) | (uint64_t)(lane * 4 + u);
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) s_best = (int)(best & 0xffffffffull) * ROWS;
    }
    __syncthreads();
    const int brow = s_best;
    const unsigned h = (unsigned)i * 2654435761u + (unsigned)g * 40503u;
    const int d0 = (h >> 4) % P, d1 = (h >> 16) % P;
    double x[4], s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * LPR + l;
        if (!OWN) xi[q] = src[(size_t)i * N + e];
        const double a = src[(size_t)d0 * N + e], b = src[(size_t)d1 * N + e], gb = src[(size_t)brow * N + e];
        const double u = ((h >> q) & 1) ? 0.25 * (gb + xi[q]) + 0.5 * (a - b) : xi[q];
        x[q] = u;
        s += u * u;
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (NT)
            __builtin_nontemporal_store(x[q], dst + (size_t)i * N + q * LPR + l);
        else
            dst[(size_t)i * N + q * LPR + l] = x[q];
    }
    if (l == 0) s_val[r] = s;
    __syncthreads();
    if (tid == 0) {
        float m = (float)s_val[0];
        for (int kk = 1; kk < ROWS; ++kk) m = fminf(m, (float)s_val[kk]);
        rec[(size_t)(par ^ 1) * WG + wg] = ((uint64_t)(uint32_t)(g + 1) << 32) | (uint64_t)__float_as_uint(m);
    }
}
__global__ void bump(int *gen_p, int by) { *gen_p += by; }

static double checksum(const double *d, std::vector<double> &h) {
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0.0;
    for (size_t k = 0; k < h.size(); ++k) s += h[k] * (double)((k % 97) + 1);
    return s;
}

int main() {
    double *A, *B;
    uint64_t *rec;
    int *err, *gen_p;
    Stamps *st;
    CK(hipMalloc(&A, (size_t)P * N * 8));
    CK(hipMalloc(&B, (size_t)P * N * 8));
    CK(hipMalloc(&rec, 2 * WG * 8));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&gen_p, 4));
    CK(hipMalloc(&st, sizeof(Stamps)));
    std::vector<double> h((size_t)P * N), h0((size_t)P * N);
    for (size_t k = 0; k < h0.size(); ++k) h0[k] = (double)((k * 2654435761u) % 1000) / 500.0 - 1.0;
    std::vector<uint64_t> r0(2 * WG);
    for (int k = 0; k < WG; ++k) r0[k] = ((uint64_t)0u << 32) | (uint64_t)0x3f800000u + (uint64_t)k, r0[WG + k] = ~0ull;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int gens = 4000;
    const long long timeout = 200000000ll;  // 2 s of the 100 MHz wall clock
    auto reset = [&]() {
        CK(hipMemcpy(A, h0.data(), h0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(B, 0, h0.size() * 8));
        CK(hipMemcpy(rec, r0.data(), r0.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(gen_p, 0, 4));
        CK(hipMemset(st, 0, sizeof(Stamps)));
    };
    // ---- per-launch form: a graph of 50 generations, replayed
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipGraphCreate(&graph, 0));
        hipGraphNode_t prev = nullptr;
        const int chunk = 50;
        for (int k = 0; k <= chunk; ++k) {
            hipKernelNodeParams kp = {};
            int kk = k, by = chunk;
            void *a1[] = {&A, &B, &rec, &gen_p, &kk, &err};
            void *a2[] = {&gen_p, &by};
            kp.func = k < chunk ? (void *)one_generation : (void *)bump;
            kp.gridDim = dim3(k < chunk ? WG : 1);
            kp.blockDim = dim3(k < chunk ? THREADS : 1);
            kp.kernelParams = k < chunk ? a1 : a2;
            hipGraphNode_t node;
            CK(hipGraphAddKernelNode(&node, graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            reset();
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < gens / chunk; ++k) CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep)
                printf("%-78s %6.2f us per generation   checksum %.17g\n",
                       "one launch per generation (graph of 50 + 1 counter kernel, replayed)", ms * 1e3 / gens,
                       checksum((gens & 1) ? B : A, h));
        }
    }
    // ---- the per-launch form with streaming stores / the own row requested early
    struct { const char *name; const void *fn; } lv[] = {
        {"one launch per generation, streaming (nt) row stores", (const void *)one_generation_v<false, true>},
        {"one launch per generation, own row requested before the generation number", (const void *)one_generation_v<true, false>},
        {"one launch per generation, both", (const void *)one_generation_v<true, true>},
    };
    for (auto &c : lv) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipGraphCreate(&graph, 0));
        hipGraphNode_t prev = nullptr;
        const int chunk = 50;
        for (int k = 0; k <= chunk; ++k) {
            hipKernelNodeParams kp = {};
            int kk = k, by = chunk;
            void *a1[] = {&A, &B, &rec, &gen_p, &kk, &err};
            void *a2[] = {&gen_p, &by};
            kp.func = k < chunk ? (void *)c.fn : (void *)bump;
            kp.gridDim = dim3(k < chunk ? WG : 1);
            kp.blockDim = dim3(k < chunk ? THREADS : 1);
            kp.kernelParams = k < chunk ? a1 : a2;
            hipGraphNode_t node;
            CK(hipGraphAddKernelNode(&node, graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            reset();
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < gens / chunk; ++k) CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-78s %6.2f us per generation   checksum %.17g\n", c.name, ms * 1e3 / gens, checksum((gens & 1) ? B : A, h));
        }
    }
    // ---- persistent forms
    struct { const char *name; int v; } cfg[] = {
        {"persistent, F: plain stores + release fence -> record -> acquire fence + plain loads", FENCE},
        {"persistent, W: write-through (sc1) stores -> record -> acquire fence + plain loads", WTHRU},
        {"persistent, S: system-scope (sc0 sc1) stores and loads, no fences", SYSTEM},
    };
    for (auto &c : cfg) {
        for (int rep = 0; rep < 2; ++rep) {
            reset();
            int g0 = 0, gg = gens, sg = gens / 2;
            void *args[] = {&A, &B, &rec, &g0, &gg, &err, (void *)&timeout, &st, &sg};
            CK(hipEventRecord(e0, s));
            const void *fn = c.v == FENCE ? (const void *)persistent<FENCE>
                                          : c.v == WTHRU ? (const void *)persistent<WTHRU> : (const void *)persistent<SYSTEM>;
            CK(hipLaunchKernel(fn, dim3(WG), dim3(THREADS), args, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int herr;
            Stamps hs;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost));
            if (rep) {
                printf("%-78s %6.2f us per generation   checksum %.17g%s\n", c.name, ms * 1e3 / gens,
                       checksum((gens & 1) ? B : A, h), herr ? "   TIMED OUT" : "");
                printf("    workgroup 0, generation %d, shader cycles: wait for the 256 records %llu, acquire + barrier %llu, row loads %llu, "
                       "stores (+ drain) %llu, barrier + release + record %llu\n",
                       sg, hs.t[1] - hs.t[0], hs.t[2] - hs.t[1], hs.t[3] - hs.t[2], hs.t[4] - hs.t[3], hs.t[5] - hs.t[4]);
            }
        }
    }
    return 0;
}
