// How long does the FIRST load of a kernel take after a kernel boundary, for (a) a word the previous kernel's workgroup 0
// wrote (the chained DE kernel's state word) and (b) a word nobody has written since before the graph started?
// 50 dependent kernels in one hipGraph, replayed; per-workgroup latency in 10 ns ticks (s_memrealtime).
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 boundary_read_probe.cpp -o /tmp/brp && /tmp/brp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(long long *state, const long long *constw, unsigned *lat, int which, int node, int nwg) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        long long v = which == 0 ? __atomic_load_n(state, __ATOMIC_RELAXED) : __atomic_load_n(constw, __ATOMIC_RELAXED);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(v) : : "memory");  // the value has arrived
        const unsigned long long t1 = wall_clock64() + (v == 0x7fffffffffffffffll ? 1 : 0);
        lat[node * nwg + blockIdx.x] = (unsigned)(t1 - t0);
    }
    // some work so that kernels are not back-to-back empties, then workgroup 0 publishes a new state word
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) state[0] = node + 1;
}

int main() {
    const int nwg = 256, nodes = 50;
    long long *state, *constw;
    unsigned *lat;
    CK(hipMalloc(&state, 64)); CK(hipMalloc(&constw, 64)); CK(hipMalloc(&lat, sizeof(unsigned) * nodes * nwg));
    CK(hipMemset(state, 0, 64)); CK(hipMemset(constw, 0, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int which = 0; which < 2; ++which) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, st, state, constw, lat, which, i, nwg);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 5; ++rep) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(nodes * nwg);
        CK(hipMemcpy(h.data(), lat, sizeof(unsigned) * nodes * nwg, hipMemcpyDeviceToHost));
        std::vector<unsigned> v(h.begin() + nwg, h.end());  // skip the first node
        std::sort(v.begin(), v.end());
        printf("%s: first-load latency per workgroup over %d kernels x %d workgroups: p10 %u0 ns, median %u0 ns, p90 %u0 ns\n",
               which == 0 ? "word written by the previous kernel (state)" : "word constant during the graph            ", nodes - 1, nwg,
               v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
    return 0;
}
