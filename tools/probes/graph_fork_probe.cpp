// Does a forked hipGraph branch run beside the main chain, and what do the fork / join edges cost?  (round 3, CPSO:
// the restart's threshold selection -- one workgroup, ~11 us -- depends on the generation kernel only, not on the
// best/termination and radius kernels that follow it.)
// Per iteration:  A (2048 workgroups, ~30 us of streaming) -> B (1 wg, ~5 us) -> C (2048 wgs, ~7 us) -> D (1 wg, short)
//                 E (1 wg, ~11 us of dependent LDS work): linear = after C, forked = A -> E -> D beside B -> C.
// build: hipcc -O3 --offload-arch=gfx950 graph_fork_probe.cpp -o graph_fork_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void stream_kernel(double *x, long n, int reps) {  // A / C: read-modify-write of a slab
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r)
        for (long k = i; k < n; k += (long)gridDim.x * blockDim.x) x[k] = x[k] * 1.0000001 + 1e-9;
}
__global__ void spin_kernel(double *out, int iters) {  // B / D / E: one workgroup, a dependent chain
    __shared__ double s[1024];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double v = 0.0;
    for (int i = 0; i < iters; ++i) {
        v += s[(threadIdx.x + i) & 1023];
        __syncthreads();
        s[threadIdx.x] = v * 1e-3;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = v;
}
static int add(hipGraph_t g, hipGraphNode_t *node, std::vector<hipGraphNode_t> deps, void *fn, dim3 grid, dim3 blk, void **args) {
    hipKernelNodeParams kp = {};
    kp.func = fn, kp.gridDim = grid, kp.blockDim = blk, kp.kernelParams = args;
    CK(hipGraphAddKernelNode(node, g, deps.data(), deps.size(), &kp));
    return 0;
}
int main() {
    const long nA = 4l << 20, nC = 1l << 20;  // doubles: 32 MB / 8 MB slabs
    double *xa, *xc, *o;
    CK(hipMalloc(&xa, nA * 8)); CK(hipMalloc(&xc, nC * 8)); CK(hipMalloc(&o, 64));
    CK(hipMemset(xa, 0, nA * 8)); CK(hipMemset(xc, 0, nC * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    long na = nA, nc = nC; int one = 1, itB = 60, itD = 8, itE = 160;
    double *oB = o, *oD = o + 1, *oE = o + 2;
    void *aA[] = {&xa, &na, &one}, *aC[] = {&xc, &nc, &one}, *aB[] = {&oB, &itB}, *aD[] = {&oD, &itD}, *aE[] = {&oE, &itE};
    const int ITER = 50;
    for (int mode = 0; mode < 3; ++mode) {  // 0: without E; 1: linear; 2: forked
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        for (int i = 0; i < ITER; ++i) {
            hipGraphNode_t A, B, C, D, E;
            std::vector<hipGraphNode_t> dep; if (prev) dep.push_back(prev);
            if (add(g, &A, dep, (void *)stream_kernel, dim3(2048), dim3(512), aA)) return 1;
            if (add(g, &B, {A}, (void *)spin_kernel, dim3(1), dim3(1024), aB)) return 1;
            if (add(g, &C, {B}, (void *)stream_kernel, dim3(2048), dim3(512), aC)) return 1;
            if (mode == 0) { if (add(g, &D, {C}, (void *)spin_kernel, dim3(1), dim3(1024), aD)) return 1; }
            if (mode == 1) { if (add(g, &E, {C}, (void *)spin_kernel, dim3(1), dim3(1024), aE)) return 1;
                             if (add(g, &D, {E}, (void *)spin_kernel, dim3(1), dim3(1024), aD)) return 1; }
            if (mode == 2) { if (add(g, &E, {A}, (void *)spin_kernel, dim3(1), dim3(1024), aE)) return 1;
                             if (add(g, &D, {C, E}, (void *)spin_kernel, dim3(1), dim3(1024), aD)) return 1; }
            prev = D;
        }
        hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int REP = 20;
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per iteration\n", mode == 0 ? "A->B->C->D (no E)      " : mode == 1 ? "A->B->C->E->D (linear)  " : "A->{B->C, E}->D (forked)", ms * 1e3 / (REP * ITER));
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    return 0;
}
