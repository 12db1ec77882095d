// Probe for DESIGN.md section 4 (PSO at BASELINE config 3), follow-up of row_stream_probe.cpp: the real generation kernel
// needs 40.9 us where its memory work alone needs 25 us, because a wave loads, THEN computes (~900 vector instructions
// per row: two Philox calls, four cosines, the sums), THEN stores.  Does prefetching the NEXT row into LDS with
// direct-to-LDS loads (global_load_lds_dwordx4: no registers) while the current row is computed close that gap?
//   A: one row per wave, registers, no loop                (the shape of the shipped kernel)
//   B: persistent waves, 8 rows each, next row requested into the other LDS buffer before the current one is computed
// Same synthetic arithmetic in both (Philox4x32-10 x 2, cos x 4, a wave reduction), same loads and stores as the real
// kernel (X, V, pbest in; X, V out; pbest out for a quarter of the rows).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/row_pipe_probe.cpp -o /tmp/rpp && /tmp/rpp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 256, P = 16384;

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// the row's arithmetic: x, v, p (4 elements per lane) -> new x, v; returns the "fitness"
__device__ __forceinline__ double row_math(double (&x)[4], double (&v)[4], const double (&p)[4], const double (&g)[4], uint32_t row,
                                           int lane) {
    double r1[4], r2[4];
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
        const U4 w = philox((uint32_t)(t >> 1) * 64u + (uint32_t)lane, row, 7u, 3u, 11u, 13u);
        r1[t] = w.x * (1.0 / 4294967296.0), r2[t] = w.y * (1.0 / 4294967296.0);
        r1[t + 1] = w.z * (1.0 / 4294967296.0), r2[t + 1] = w.w * (1.0 / 4294967296.0);
    }
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[t] = (0.7 * v[t] + (1.5 * r1[t]) * (p[t] - x[t])) + (1.5 * r2[t]) * (g[t] - x[t]);
        x[t] = x[t] + v[t];
        sa += x[t] * x[t];
        sb += cos(6.283185307179586 * x[t]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sa += __shfl_xor(sa, off, 64), sb += __shfl_xor(sb, off, 64);
    return (22.718281828459045 - 20.0 * exp(-0.2 * sqrt(sa / N))) - exp(sb / N);
}

__global__ __launch_bounds__(512) void variant_a(double *X, double *V, double *B, double *F, const double *G) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    double *xr = X + row * N, *vr = V + row * N, *br = B + row * N;
    double x[4], v[4], p[4], g[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = xr[t * 64 + lane], v[t] = vr[t * 64 + lane], p[t] = br[t * 64 + lane], g[t] = G[t * 64 + lane];
    const double f = row_math(x, v, p, g, (uint32_t)row, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) vr[t * 64 + lane] = v[t], xr[t * 64 + lane] = x[t];
    if ((row & 3) == 0 || f < -1.0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) br[t * 64 + lane] = x[t];
        if (lane == 0) F[row] = f;
    }
}

// C / D / E: variant A plus what the real kernel does for bit-parity with numpy: the new positions and the two term arrays go
// through LDS (3n + 64 doubles per row = 52 KB per workgroup of 8 rows -> 3 workgroups per CU) and the sums are formed in
// numpy's pairwise order -- 16 lanes walk chains of 16 terms -- (MODE >= 1); a per-workgroup best record behind a workgroup
// barrier, reduced by thread 0 (MODE == 2), or reduced by whichever wave finishes last, no barrier (MODE == 3).
// GATE: the kernel starts, like the real one, by reading the generation state (done flag, generation counter) from
// memory and returning if the run is over -- every other load of the workgroup waits behind that round trip.
template <int MODE, bool GATE = false>
__global__ __launch_bounds__(512) void variant_c(double *X, double *V, double *B, double *F, const double *G, double *part,
                                                 const long *state = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double sf[8];
    __shared__ unsigned done;
    uint32_t gen = 7u;
    if (GATE) {
        if (state[1]) return;
        gen = (uint32_t)state[0] + 1u;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE == 3 && threadIdx.x == 0) done = 0;
    const long row = (long)blockIdx.x * 8 + wave;
    double *xr = X + row * N, *vr = V + row * N, *br = B + row * N;
    double *U = lds + wave * (3 * N + 64), *A = U + N + 8, *Bt = A + N;
    double x[4], v[4], p[4], g[4], r1[4], r2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = xr[t * 64 + lane], v[t] = vr[t * 64 + lane], p[t] = br[t * 64 + lane], g[t] = G[t * 64 + lane];
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
        const U4 w = philox((uint32_t)(t >> 1) * 64u + (uint32_t)lane, (uint32_t)row, gen, 3u, 11u, 13u);
        r1[t] = w.x * (1.0 / 4294967296.0), r2[t] = w.y * (1.0 / 4294967296.0);
        r1[t + 1] = w.z * (1.0 / 4294967296.0), r2[t + 1] = w.w * (1.0 / 4294967296.0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v[t] = (0.7 * v[t] + (1.5 * r1[t]) * (p[t] - x[t])) + (1.5 * r2[t]) * (g[t] - x[t]);
        x[t] = x[t] + v[t];
        U[t * 64 + lane] = x[t];
        vr[t * 64 + lane] = v[t], xr[t * 64 + lane] = x[t];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double xx = U[t * 64 + lane];
        A[t * 64 + lane] = xx * xx;
        Bt[t * 64 + lane] = cos(6.283185307179586 * xx);
    }
    __builtin_amdgcn_wave_barrier();
    double ca = 0.0, cb = 0.0;
    if (lane < 16) {  // two leaves of 128 terms: lane = leaf * 8 + accumulator, 16 terms each, in order
        const int leaf = lane >> 3, j = lane & 7;
#pragma unroll
        for (int h = 0; h < 16; ++h) ca += A[leaf * 128 + h * 8 + j], cb += Bt[leaf * 128 + h * 8 + j];
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) ca += __shfl_xor(ca, off, 64), cb += __shfl_xor(cb, off, 64);
    const double sa = __shfl(ca, 0, 64), sb = __shfl(cb, 0, 64);
    const double f = (22.718281828459045 - 20.0 * exp(-0.2 * sqrt(sa / N))) - exp(sb / N);
    const bool better = (row & 3) == 0 || f < -1.0;
    if (better) {
#pragma unroll
        for (int t = 0; t < 4; ++t) br[t * 64 + lane] = U[t * 64 + lane];
        if (lane == 0) F[row] = f;
    }
    if (MODE == 2) {
        if (lane == 0) sf[wave] = f;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = sf[0];
            for (int k = 1; k < 8; ++k) m = fmin(m, sf[k]);
            part[blockIdx.x] = m;
        }
    }
    if (MODE == 3) {
        if (lane == 0) {
            sf[wave] = f;
            __threadfence_block();
            if (atomicAdd(&done, 1u) == 7u) {
                double m = sf[0];
                for (int k = 1; k < 8; ++k) m = fmin(m, sf[k]);
                part[blockIdx.x] = m;
            }
        }
    }
}

#define LDS_DMA16(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src), (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)

// 4 waves per workgroup; per wave two buffers of 3 x 256 doubles
__global__ __launch_bounds__(256) void variant_b(double *X, double *V, double *B, double *F, const double *G, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *buf = lds + wave * (2 * 3 * N);
    const long w0 = (long)blockIdx.x * 4 + wave;
    double g[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = G[t * 64 + lane];
    auto request = [&](long row, double *dst) {  // 6 instructions, 1 KB each, no registers
        const double *xr = X + row * N, *vr = V + row * N, *br = B + row * N;
        LDS_DMA16(xr + lane * 2, dst);
        LDS_DMA16(xr + 128 + lane * 2, dst + 128);
        LDS_DMA16(vr + lane * 2, dst + N);
        LDS_DMA16(vr + 128 + lane * 2, dst + N + 128);
        LDS_DMA16(br + lane * 2, dst + 2 * N);
        LDS_DMA16(br + 128 + lane * 2, dst + 2 * N + 128);
    };
    long row = w0;
    int cur = 0;
    if (row < P) request(row, buf);
    for (; row < P; row += nwaves, cur ^= 1) {
        double *mine = buf + cur * (3 * N);
        const long nxt = row + nwaves;
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): this row has landed (and the previous row's stores are out)
        if (nxt < P) request(nxt, buf + (cur ^ 1) * (3 * N));
        double x[4], v[4], p[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = mine[t * 64 + lane], v[t] = mine[N + t * 64 + lane], p[t] = mine[2 * N + t * 64 + lane];
        const double f = row_math(x, v, p, g, (uint32_t)row, lane);
        double *xr = X + row * N, *vr = V + row * N, *br = B + row * N;
#pragma unroll
        for (int t = 0; t < 4; ++t) vr[t * 64 + lane] = v[t], xr[t * 64 + lane] = x[t];
        if ((row & 3) == 0 || f < -1.0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) br[t * 64 + lane] = x[t];
            if (lane == 0) F[row] = f;
        }
    }
}

template <class Fn>
double timed(Fn launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best * 1e3 / 200;
}

int main() {
    double *X, *V, *B, *F, *G;
    const size_t bytes = (size_t)P * N * 8;
    CK(hipMalloc(&X, bytes)); CK(hipMalloc(&V, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&F, P * 8)); CK(hipMalloc(&G, N * 8));
    CK(hipMemset(X, 0, bytes)); CK(hipMemset(V, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(G, 0, N * 8));
    const double a = timed([&] { hipLaunchKernelGGL(variant_a, dim3(P / 8), dim3(512), 0, 0, X, V, B, F, G); });
    printf("A  one row per wave, registers:                      %6.2f us per launch\n", a);
    double *part;
    CK(hipMalloc(&part, (P / 8) * 8));
    const size_t ldsc = 8 * (3 * N + 64) * 8;
    printf("C  A + terms staged in LDS, sums in numpy's order:   %6.2f us per launch\n",
           timed([&] { hipLaunchKernelGGL(variant_c<1>, dim3(P / 8), dim3(512), ldsc, 0, X, V, B, F, G, part); }));
    printf("D  C + per-workgroup record behind a barrier:        %6.2f us per launch\n",
           timed([&] { hipLaunchKernelGGL(variant_c<2>, dim3(P / 8), dim3(512), ldsc, 0, X, V, B, F, G, part); }));
    printf("E  C + record by the wave that finishes last:        %6.2f us per launch\n",
           timed([&] { hipLaunchKernelGGL(variant_c<3>, dim3(P / 8), dim3(512), ldsc, 0, X, V, B, F, G, part); }));
    CK(hipFuncSetAttribute((const void *)variant_c<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
    for (size_t pad : {(size_t)0, (size_t)(54 << 10) - ldsc, (size_t)(81 << 10) - ldsc})
        printf("D  with %3zu KB of LDS per workgroup (%d workgroups = %2d waves per CU):  %6.2f us per launch\n", (ldsc + pad) >> 10,
               (int)((160 << 10) / (ldsc + pad)), 8 * (int)((160 << 10) / (ldsc + pad)),
               timed([&] { hipLaunchKernelGGL(variant_c<2>, dim3(P / 8), dim3(512), ldsc + pad, 0, X, V, B, F, G, part); }));
    long *state;
    CK(hipMalloc(&state, 64)); CK(hipMemset(state, 0, 64));
    printf("F  D + the state read that gates every other load:  %6.2f us per launch\n",
           timed([&] { hipLaunchKernelGGL((variant_c<2, true>), dim3(P / 8), dim3(512), ldsc, 0, X, V, B, F, G, part, state); }));
    for (int wgs_per_cu : {2, 3, 4, 6}) {
        const int wgs = 256 * wgs_per_cu, nwaves = wgs * 4;
        const size_t lds = 4 * 2 * 3 * N * 8;
        const double b = timed([&] { hipLaunchKernelGGL(variant_b, dim3(wgs), dim3(256), lds, 0, X, V, B, F, G, nwaves); });
        printf("B  persistent, LDS prefetch, %d workgroups of 4 waves per CU (%4.1f rows per wave): %6.2f us per launch\n", wgs_per_cu,
               (double)P / nwaves, b);
    }
    return 0;
}
