// Wide rows (n > kWideFrom, sx_device.hpp): one workgroup per individual, the summation plan in device memory (sx_wide.hip).
// The narrow entry points (sx_eval, sx_de_generation, sx_de_graph_create, sx_pso_generation, sx_pso_graph_create, ...)
// branch here on is_wide(n); records are one per row (sx_num_partials(P, n) = P).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stochopy_hip.h"
#include "sx_device.hpp"

namespace sx {

constexpr int kWideMaxDim = 262144;  // leaf sums of a row (2 (n/64 + 2) doubles) share the LDS with the stage

inline bool is_wide(int n) { return n > wide_from(); }

int wide_eval(int fun_id, const double *X, int64_t P, int n, int64_t ldx, const double *xm, const double *xstd, double *f,
              double *part_f, int64_t *part_i, int clip, const double *pen_v, double *pen_out, hipStream_t s);
int wide_warm_plan(int fun_id, int n, hipStream_t s);
int wide_de_launch(const sx_de_args *a, hipStream_t s);
int wide_de_add_node(hipGraph_t graph, hipGraphNode_t *prev, const sx_de_args *a);
int wide_pso_launch(const sx_pso_args *a, hipStream_t s);
// VD-CMA candidates [row0, row0 + rows) of generation `gen`: normals, steps y, candidates x, objective, t_k (tk_out may be NULL)
int wide_vd_candidates(const sx_vd_args *a, int64_t gen, int64_t row0, int64_t rows, double *ary_out, double *arx_out,
                       double *fit_out, double *tk_out, hipStream_t s);
int wide_pso_add_node(hipGraph_t graph, hipGraphNode_t *prev, const sx_pso_args *a);

}  // namespace sx
