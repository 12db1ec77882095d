#include <hip/hip_runtime.h>
#include <stdint.h>
struct Big { const double* a[20]; int64_t P; int n; };
__global__ void k(const int64_t *state, const double *rec, Big b, int q, double *o) {
    int64_t it = state[1];
    o[threadIdx.x] = rec[threadIdx.x] + (double)it + b.a[3][threadIdx.x] + q;
}
