// Probe for DESIGN.md section 4 (round 5, "M at P = 2^20"): what does the memory system give for the DE generation's ACCESS
// PATTERN at the metric's row length when nothing is launch-bound?  Per individual: its own row (sequential), two donor rows
// (random rows of the same 1 GiB buffer), the best row (one hot row), one row written (sequential) -- 4 112 algorithmic bytes,
// no arithmetic beyond one add per element, one wavefront per two rows of 128 doubles exactly as de_generation_kernel maps
// them (32 lanes per row, lane l owns elements 32 q + l).
//   A: donors = random rows (a hash of the row number), 8 bytes per lane and load   (the product kernel's pattern)
//   B: the same with 16 bytes per lane and load (lane owns element pairs)
//   C: donors = the next two rows (no gather: what the pattern costs without the randomness)
//   D: own row + store only (a copy: the streaming ceiling of this mapping)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/de_gather_probe.cpp -o /tmp/dgp && /tmp/dgp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 128;

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // a cheap integer hash: the donor "draw"
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(512) void gen(const double *__restrict__ cur, double *__restrict__ nxt, const double *__restrict__ best,
                                           uint32_t P, uint32_t salt) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const uint32_t row = (blockIdx.x * 8u + (threadIdx.x >> 6)) * 2u + (lane >> 5);
    uint32_t d0 = row + 1 < P ? row + 1 : 0, d1 = row + 2 < P ? row + 2 : 1;
    if (MODE == 0 || MODE == 1) d0 = mix(row ^ salt) % P, d1 = mix(row * 2654435761u + salt) % P;
    const double *x = cur + (size_t)row * N, *a = cur + (size_t)d0 * N, *b = cur + (size_t)d1 * N;
    double *o = nxt + (size_t)row * N;
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = 64 * t + 2 * l;
            const double2 xv = *(const double2 *)(x + e), av = *(const double2 *)(a + e), bv = *(const double2 *)(b + e),
                          gv = *(const double2 *)(best + e);
            double2 r;
            r.x = gv.x + 0.5 * (av.x - bv.x) + 1e-300 * xv.x, r.y = gv.y + 0.5 * (av.y - bv.y) + 1e-300 * xv.y;
            __builtin_nontemporal_store(r.x, o + e), __builtin_nontemporal_store(r.y, o + e + 1);
        }
    } else {
        double xv[4], av[4], bv[4], gv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 32 * t + l;
            xv[t] = x[e];
            if (MODE != 3) av[t] = a[e], bv[t] = b[e], gv[t] = best[e];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double r = MODE == 3 ? xv[t] : gv[t] + 0.5 * (av[t] - bv[t]) + 1e-300 * xv[t];
            __builtin_nontemporal_store(r, o + 32 * t + l);
        }
    }
}

template <int MODE>
static void run(const char *label, double *A, double *B, double *best, uint32_t P, double bytes_per_row) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = P / 16;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gen<MODE>, dim3(blocks), dim3(512), 0, 0, A, B, best, P, 17u + w);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(gen<MODE>, dim3(blocks), dim3(512), 0, 0, (r & 1) ? B : A, (r & 1) ? A : B, best, P, 1000u + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-64s P=%8u: %8.1f us  %7.1f GB/s on %4.0f B per row = %.3f of 8 TB/s\n", label, P, us, bytes_per_row * P / us / 1e3, bytes_per_row,
           bytes_per_row * P / us / 1e3 / 8000.0);
}

int main() {
    for (uint32_t P : {1u << 20, 1u << 16, 1u << 12}) {
        double *A, *B, *best;
        CK(hipMalloc(&A, (size_t)P * N * 8)); CK(hipMalloc(&B, (size_t)P * N * 8)); CK(hipMalloc(&best, N * 8));
        CK(hipMemset(A, 0, (size_t)P * N * 8)); CK(hipMemset(B, 0, (size_t)P * N * 8)); CK(hipMemset(best, 0, N * 8));
        run<0>("A  two random donor rows, 8 B per lane (the product's pattern)", A, B, best, P, 4112);
        run<1>("B  two random donor rows, 16 B per lane", A, B, best, P, 4112);
        run<2>("C  donors = the next two rows (no gather)", A, B, best, P, 4112);
        run<3>("D  own row in, row out (copy)", A, B, best, P, 2048);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(best));
    }
    return 0;
}
