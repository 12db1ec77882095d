// Probe for DESIGN.md section 4 ("M shape"): can the metric workload (DE, n=128, P=4096: two 4 MiB row buffers) run
// faster as ONE persistent kernel confined to ONE XCD (32 CUs sharing one 4 MiB L2, barrier that never leaves the
// XCD) than as one kernel per generation on all 256 CUs (10.2 us per generation)?
// The kernel does the generation's MEMORY work only (row i, two random donor rows and the best row in, the new row out,
// a cheap sum as "objective") -- a lower bound on what the real kernel could reach in this structure.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_probe.cpp -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 128, P = 4096, LPR = 32;

// COH = 0: loads past the L1 only (sc1) and plain stores -- coherent INSIDE one XCD (one L2), NOT across XCDs (each
//          XCD has its own L2): for more than one XCD the timing is a floor, the data may be stale.
// COH = 1: system-scope loads and stores (sc0 sc1: past every cache) -- what a chip-wide persistent kernel needs.
template <int COH>
__device__ __forceinline__ double ld_row(const double *p) {
    return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
               : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int COH>
__device__ __forceinline__ void st_row(double *p, double v) {
    if (COH)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        *p = v;
}

// active workgroups: those with blockIdx.x % stride == 0 (stride 8 -> the 32 workgroups the dispatcher puts on XCD 0)
template <int COH>
__global__ __launch_bounds__(1024) void probe(double *A, double *B, double *fit, const double *best, unsigned *counter,
                                              int gens, int stride, int nactive, unsigned *xcc_seen) {
    if (blockIdx.x % stride) return;
    const int wg = blockIdx.x / stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & (LPR - 1), sub = lane / LPR;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&xcc_seen[0], 1u << (xcc & 15));
    }
    const int rows_per_wg = P / nactive;  // 128 with 32 workgroups, 16 with 256
    for (int g = 0; g < gens; ++g) {
        const double *src = (g & 1) ? B : A;
        double *dst = (g & 1) ? A : B;
        for (int r = wave * 2 + sub; r < rows_per_wg; r += 32) {
            const int i = wg * rows_per_wg + r;
            unsigned h = (unsigned)i * 2654435761u + (unsigned)g * 40503u;
            const int d0 = (h >> 4) % P, d1 = (h >> 16) % P;
            double s = 0.0, x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = q * LPR + l;
                const double xi = ld_row<COH>(src + (size_t)i * N + e), a = ld_row<COH>(src + (size_t)d0 * N + e),
                             b = ld_row<COH>(src + (size_t)d1 * N + e);
                const double u = ((h >> q) & 1) ? best[e] + 0.5 * (a - b) : xi;
                x[q] = u;
                s += u * u;
            }
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
            const bool better = s < fit[i] || true;  // always write: the traffic of a winning row
#pragma unroll
            for (int q = 0; q < 4; ++q) st_row<COH>(dst + (size_t)i * N + q * LPR + l, better ? x[q] : 0.0);
            if (l == 0) fit[i] = s;
        }
        // barrier over the active workgroups: stores drained, one arrival per workgroup, poll past the caches
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nactive * (unsigned)(g + 1);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main() {
    double *A, *B, *fit, *best;
    unsigned *counter, *xcc;
    CK(hipMalloc(&A, (size_t)P * N * 8));
    CK(hipMalloc(&B, (size_t)P * N * 8));
    CK(hipMalloc(&fit, P * 8));
    CK(hipMalloc(&best, N * 8));
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&xcc, 4));
    std::vector<double> h((size_t)P * N);
    for (size_t k = 0; k < h.size(); ++k) h[k] = (double)((k * 2654435761u) % 1000) / 500.0 - 1.0;
    CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(best, h.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemset(fit, 0x7f, P * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int gens = 2000;
    struct { const char *name; int stride, nactive, coh; } cfg[] = {
        {"one XCD: 32 workgroups (blockIdx % 8 == 0), intra-XCD barrier", 8, 32, 0},
        {"two XCDs, L2-only accesses (floor, not coherent)", 4, 64, 0},
        {"all 8 XCDs, L2-only accesses (floor, not coherent)", 1, 256, 0},
        {"one XCD, system-scope loads/stores", 8, 32, 1},
        {"all 8 XCDs, system-scope loads/stores (coherent)", 1, 256, 1},
    };
    for (auto &c : cfg) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4));
            CK(hipMemset(xcc, 0, 4));
            CK(hipEventRecord(e0));
            if (c.coh)
                hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 0, 0, A, B, fit, best, counter, gens, c.stride, c.nactive, xcc);
            else
                hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 0, 0, A, B, fit, best, counter, gens, c.stride, c.nactive, xcc);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned seen;
            CK(hipMemcpy(&seen, xcc, 4, hipMemcpyDeviceToHost));
            if (rep) printf("%-66s %7.2f us per generation (XCC ids seen: 0x%02x)\n", c.name, ms * 1e3 / gens, seen);
        }
    }
    return 0;
}
