// Probe for DESIGN.md section 4 (PSO at BASELINE config 3): what does the access WIDTH buy on this access pattern?
// The PSO generation streams three row arrays in (X, V, pbest) and three out (X, V, pbest where improved), one wavefront
// per row of n = 256 doubles, 8 rows per workgroup, 8 bytes per lane and instruction (lane l owns elements l, l+64, ...).
// The round-1 review proposed 16 bytes per lane (lane owns element pairs).  This kernel does the generation's MEMORY
// work only -- same grid, same rows, a trivial update -- with W = 1 (global_load_dwordx2) or W = 2 (dwordx4) doubles
// per lane and instruction, and with all / a quarter of the pbest rows written.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/row_stream_probe.cpp -o /tmp/rsp && /tmp/rsp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 256, P = 16384, ROWS = 8;

template <int W>
__global__ __launch_bounds__(ROWS * 64) void stream(double *X, double *V, double *B, const double *g, int every) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * ROWS + wave;
    double *x = X + row * N, *v = V + row * N, *b = B + row * N;
    constexpr int STEPS = N / (64 * W);
    double xv[STEPS][W], vv[STEPS][W], bv[STEPS][W], gv[STEPS][W];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int e = (s * 64 + lane) * W;
        if (W == 2) {
            const double2 a = *(const double2 *)(x + e), c = *(const double2 *)(v + e), d = *(const double2 *)(b + e),
                          h = *(const double2 *)(g + e);
            xv[s][0] = a.x, xv[s][W - 1] = a.y, vv[s][0] = c.x, vv[s][W - 1] = c.y, bv[s][0] = d.x, bv[s][W - 1] = d.y;
            gv[s][0] = h.x, gv[s][W - 1] = h.y;
        } else {
            xv[s][0] = x[e], vv[s][0] = v[e], bv[s][0] = b[e], gv[s][0] = g[e];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int w = 0; w < W; ++w) {
            vv[s][w] = 0.7 * vv[s][w] + 0.3 * (bv[s][w] - xv[s][w]) + 0.2 * (gv[s][w] - xv[s][w]);
            xv[s][w] = xv[s][w] + vv[s][w];
            acc += xv[s][w] * xv[s][w];
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const bool better = (row % every) == 0 || acc < 0.0;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int e = (s * 64 + lane) * W;
        if (W == 2) {
            *(double2 *)(v + e) = make_double2(vv[s][0], vv[s][W - 1]);
            *(double2 *)(x + e) = make_double2(xv[s][0], xv[s][W - 1]);
            if (better) *(double2 *)(b + e) = make_double2(xv[s][0], xv[s][W - 1]);
        } else {
            v[e] = vv[s][0];
            x[e] = xv[s][0];
            if (better) b[e] = xv[s][0];
        }
    }
}

template <int W>
void run(double *X, double *V, double *B, double *g, int every) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stream<W>, dim3(P / ROWS), dim3(ROWS * 64), 0, 0, X, V, B, g, every);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(stream<W>, dim3(P / ROWS), dim3(ROWS * 64), 0, 0, X, V, B, g, every);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double us = best * 1e3 / 200;
    const double bytes = (3.0 + 2.0 + 1.0 / every) * P * N * 8.0;
    printf("%2d bytes per lane, pbest written for 1 row in %d: %6.2f us per launch (back to back) -> %6.0f GB/s of %5.1f MB\n", 8 * W, every,
           us, bytes / us / 1e3, bytes / 1e6);
}

int main() {
    double *X, *V, *B, *g;
    const size_t bytes = (size_t)P * N * 8;
    CK(hipMalloc(&X, bytes)); CK(hipMalloc(&V, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&g, N * 8));
    CK(hipMemset(X, 0, bytes)); CK(hipMemset(V, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(g, 0, N * 8));
    for (int every : {1, 4}) {
        run<1>(X, V, B, g, every);
        run<2>(X, V, B, g, every);
    }
    return 0;
}
