// Row-level device routines shared by the evaluation / DE / PSO kernels:
// wave-per-individual indexing, LDS-staged objective evaluation, per-workgroup best.
#pragma once
#include "sx_device.hpp"

namespace sx {

struct RowIds {
    int wave, lane;
    int64_t row, rowc;
    bool active;
    __device__ __forceinline__ explicit RowIds(int64_t P) {
        wave = (int)(threadIdx.x >> 6);
        lane = (int)(threadIdx.x & 63);
        const int rpb = (int)(blockDim.x >> 6);
        row = (int64_t)blockIdx.x * rpb + wave;
        active = row < P;
        rowc = active ? row : P - 1;  // padding waves shadow the last row and store nothing
    }
};

// LDS traffic of one wavefront is processed in issue order, so data a lane wrote is
// visible to the other lanes of the SAME wave once the write has been issued; this
// only keeps the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void lds_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Objective of the row staged in LDS at U[0..n): terms by all 64 lanes -> A/B (behind U),
// then the numpy-order row sums.  Each wave works on its own LDS slice (no workgroup barrier).
template <int FUN>
__device__ __forceinline__ double row_objective(double *U, int n, const PlanArg &plan, int lane) {
    using O = Obj<FUN>;
    double *A = U + n + 8;
    double *B = A + n;
    const int m = O::NEXT ? n - 1 : n;
    lds_wave_fence();  // U complete (written and read by this wave only)
    for (int e = lane; e < m; e += kWave) {
        const double x = U[e];
        const double xn = O::NEXT ? U[e + 1] : 0.0;
        double a, b;
        O::term(x, xn, e, a, b);
        A[e] = a;
        if (O::TWO) B[e] = b;
    }
    lds_wave_fence();  // terms complete
    double sa, sb;
    row_reduce2<O::TWO, O::BMUL>(A, B, B + n, plan, lane, sa, sb);
    return O::finish(sa, sb, n);
}

// one (min f, first row) record per workgroup for the best-of-generation kernel
__device__ __forceinline__ void block_partial(double val, const RowIds &id, double *sf, int64_t *si,
                                              double *__restrict__ part_f, int64_t *__restrict__ part_i) {
    if (id.lane == 0) {
        sf[id.wave] = id.active ? val : __builtin_huge_val();
        si[id.wave] = id.active ? id.row : INT64_MAX;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bf = sf[0];
        int64_t bi = si[0];
        const int rpb = (int)(blockDim.x >> 6);
        for (int k = 1; k < rpb; ++k) argmin_combine(bf, bi, sf[k], si[k]);
        part_f[blockIdx.x] = bf;
        part_i[blockIdx.x] = bi;
    }
}

}  // namespace sx
