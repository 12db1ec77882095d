// Differential Evolution: one fused kernel per generation
// (donor gather -> mutation -> binomial crossover -> bound repair -> objective
//  -> greedy selection -> per-workgroup best), plus the hipGraph of generations.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/de/_de.py:304-311   delete_shuffle_sync (donor indices)
//   stochopy/optimize/de/_de.py:314-351   de_sync
//   stochopy/optimize/de/_strategy.py:1-38  rand1bin / rand2bin / best1bin / best2bin
//   stochopy/optimize/de/_constraints.py:13-28  Random
//   stochopy/optimize/_common.py:123-130  selection (strict <, in place)
//   stochopy/factory/benchmark.py         objective, fused
//
// 16/32/64 lanes per individual (sx_rowops.hpp); the population is double-buffered: generation g
// lives in buf[g & 1], its successor is written to the other buffer (winner or
// unchanged row), so donor reads never race with selection writes.
#define SX_DE_XM 0
#include "sx_de_kernel.hpp"
#include "sx_wide.hpp"

namespace sx {
// the chained (sx_de_chain.hip) and peer-exchange (sx_de_p2p.hip) kernels, as launchable function pointers
void *de_chain_kernel(int fun_id, int n, int64_t P, int strategy, int constraints);
void *de_p2p_kernel(int fun_id, int n, int64_t P, int strategy, int constraints);
}

namespace {

int check_args(const sx_de_args *a) {
    SX_REQUIRE(a != nullptr, "sx_de: null args");
    SX_REQUIRE(a->buf0 && a->buf1 && a->fit && a->state && a->part_f && a->part_i, "sx_de: null device pointer");
    SX_REQUIRE(a->P >= 2 && a->P < (int64_t)1 << 31 && a->n >= 1 && a->ld >= a->n, "sx_de: bad shape");
    SX_REQUIRE(a->fun_id >= 0 && a->fun_id < SX_FUN_COUNT, "sx_de: unknown objective");
    SX_REQUIRE(a->strategy >= 0 && a->strategy <= 3, "sx_de: unknown strategy");
    SX_REQUIRE(a->P - 1 >= donors_of(a->strategy), "sx_de: popsize too small for the strategy");
    SX_REQUIRE(a->rng == SX_RNG_HOST || a->rng == SX_RNG_PHILOX, "sx_de: unknown rng mode");
    if (a->rng == SX_RNG_HOST) {
        SX_REQUIRE(a->r1 && a->donors && a->irand, "sx_de: host draws missing");
        SX_REQUIRE(a->constraints == 0 || a->resample, "sx_de: resample block missing");
    }
    SX_REQUIRE(a->constraints == 0 || (a->lower && a->upper), "sx_de: bounds missing");
    return 0;
}

de_kernel_t kernel_for(const sx_de_args *a) {
    return a->rng == SX_RNG_PHILOX ? pick_kernel<SX_RNG_PHILOX, 0>(a->fun_id, a->n, a->P, a->strategy, a->constraints)
                                   : pick_kernel<SX_RNG_HOST, 0>(a->fun_id, a->n, a->P, a->strategy, a->constraints);
}

Geometry geometry(const sx_de_args *a) {
    Geometry g = row_geometry(a->P, a->n);
    // rows of up to 256 elements stage nothing but the trial vector (sx_device.hpp gen_row_stride)
    if (!is_wide(a->n)) g.lds = (size_t)rows_per_block(a->n) * gen_row_stride(a->n) * sizeof(double);
    return g;
}

}  // namespace

extern "C" int sx_de_generation(const sx_de_args *a, int finalize, void *stream) {
    if (int rc = check_args(a)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const Geometry g = geometry(a);
    if (is_wide(a->n)) {  // rows of more than sx_wide_from() elements: one workgroup per row (sx_wide.hip), one record per row
        if (int rc = wide_de_launch(a, s)) return rc;
    } else {
        PlanArg plan;
        if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
        hipLaunchKernelGGL(kernel_for(a), dim3(g.blocks), dim3(g.threads), g.lds, s, (const sx_state *)nullptr,
                           (const double *)nullptr, (const int64_t *)nullptr, (int64_t)g.blocks, *a, plan, 0, 0,
                           sx_xchg_args{});
        SX_LAUNCH_CHECK();
    }
    if (finalize) {
        SX_REQUIRE(a->gbest != nullptr, "sx_de_generation: the separate finalize kernel needs the gbest buffer");
        return sx_select_finalize(a->part_f, a->part_i, g.blocks, a->buf0, a->buf1, a->ld, a->n, a->gbest, a->state,
                                  a->maxiter, a->xtol, a->ftol, stream);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// hipGraph of ngen generations: 2*ngen kernel nodes in a chain, every node
// identical (per-generation state is read from a.state on the device).
// ---------------------------------------------------------------------------
extern "C" int sx_de_graph_create(const sx_de_args *a, int ngen, sx_graph **out) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(out != nullptr && ngen >= 1 && a->gbest != nullptr, "sx_de_graph_create: bad arguments");
    SX_REQUIRE(a->rng == SX_RNG_PHILOX, "sx_de_graph_create: graphs need in-kernel (Philox) draws");
    PlanArg plan = {};
    const bool wide = is_wide(a->n);
    if (!wide && make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a);
    sx_graph *gr = new sx_graph();
    SX_HIP(hipGraphCreate(&gr->graph, 0));
    sx_de_args args = *a;
    int zero = 0;
    int64_t npart = g.blocks;
    sx_xchg_args nox = {};
    const void *none = nullptr;
    void *kargs[] = {&none, &none, &none, &npart, &args, &plan, &zero, &zero, &nox};
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < ngen; ++i) {
        if (wide) {
            if (int rc = wide_de_add_node(gr->graph, &prev, a)) return rc;
        } else {
            hipKernelNodeParams kp = {};
            kp.func = (void *)kernel_for(a);
            kp.gridDim = dim3(g.blocks);
            kp.blockDim = dim3(g.threads);
            kp.sharedMemBytes = (unsigned)g.lds;
            kp.kernelParams = kargs;
            kp.extra = nullptr;
            hipGraphNode_t node;
            SX_HIP(hipGraphAddKernelNode(&node, gr->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
        if (int rc = add_finalize_node(gr->graph, &prev, a->part_f, a->part_i, g.blocks, a->buf0, a->buf1, a->ld, a->n,
                                       a->gbest, a->state, a->maxiter, a->xtol, a->ftol))
            return rc;
    }
    SX_HIP(hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0));
    *out = gr;
    return 0;
}

// Multi-GPU: this shard's generation + its best-of-generation record, one host call (the caller then
// all-gathers the records and calls sx_gather_finalize).
extern "C" int sx_de_shard_generation(const sx_de_args *a, double *record, void *stream) {
    SX_REQUIRE(record != nullptr, "sx_de_shard_generation: null record");
    if (int rc = sx_de_generation(a, 0, stream)) return rc;
    return sx_shard_best(a->part_f, a->part_i, (int64_t)geometry(a).blocks, a->buf0, a->buf1, a->ld, a->n, a->state,
                         a->row0, record, stream);
}

// ---------------------------------------------------------------------------
// Chained finalize (single GPU, Philox): one kernel per generation.
// ---------------------------------------------------------------------------
// every wavefront of the single-GPU chained kernel re-reduces the workgroup records, so only while those are
// few; in the peer-exchange kernel only the service workgroup reads them (any number)
static int check_chain(const sx_de_args *a, bool peer_exchange) {
    if (int rc = check_args(a)) return rc;
    SX_REQUIRE(a->rng == SX_RNG_PHILOX, "sx_de_chain: needs in-kernel (Philox) draws");
    SX_REQUIRE(!is_wide(a->n), "sx_de_chain: rows served by the one-workgroup-per-row kernels (n > sx_wide_from()) take the two-kernel path (sx_de_generation)");
    SX_REQUIRE(peer_exchange || sx_num_partials(a->P, a->n) <= 512,
               "sx_de_chain: more than 512 workgroup records (use the two-kernel path)");
    return 0;
}

// x == nullptr: single GPU (XM = 1); otherwise the peer-exchange kernel (XM = 2, one service workgroup in front)
static int chain_launch(const sx_de_args *a, const sx_xchg_args *x, int parity, int finalize_only, void *stream) {
    if (int rc = check_chain(a, x != nullptr)) return rc;
    SX_REQUIRE(parity == 0 || parity == 1, "sx_de_chain_launch: parity must be 0 or 1");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a);
    de_kernel_t kern = (de_kernel_t)(x ? de_p2p_kernel(a->fun_id, a->n, a->P, a->strategy, a->constraints)
                                       : de_chain_kernel(a->fun_id, a->n, a->P, a->strategy, a->constraints));
    const unsigned blocks = finalize_only ? 1u : g.blocks + (x ? 1u : 0u);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(g.threads), g.lds, (hipStream_t)stream,
                       (const sx_state *)(a->state + parity), (const double *)(a->part_f + (int64_t)parity * g.blocks),
                       (const int64_t *)(a->part_i + (int64_t)parity * g.blocks), (int64_t)g.blocks, *a, plan, parity,
                       finalize_only ? 1 : 0, x ? *x : sx_xchg_args{});
    SX_LAUNCH_CHECK();
    return 0;
}

static int chain_graph_create(const sx_de_args *a, const sx_xchg_args *x, int ngen, int start_parity, sx_graph **out) {
    if (int rc = check_chain(a, x != nullptr)) return rc;
    SX_REQUIRE(out != nullptr && ngen >= 1 && (start_parity == 0 || start_parity == 1),
               "sx_de_chain_graph_create: bad arguments");
    PlanArg plan;
    if (make_plan_arg(a->fun_id, a->n, &plan)) return -1;
    const Geometry g = geometry(a);
    sx_graph *gr = new sx_graph();
    SX_HIP(hipGraphCreate(&gr->graph, 0));
    sx_de_args args = *a;
    sx_xchg_args xa = x ? *x : sx_xchg_args{};
    int mode = 0;
    int64_t npart = g.blocks;
    hipGraphNode_t prev = nullptr;
    for (int i = 0; i < ngen; ++i) {
        int parity = (start_parity + i) & 1;
        const sx_state *sin_pre = a->state + parity;  // preloadable leading arguments (sx_de_kernel.hpp)
        const double *pf_pre = a->part_f + (int64_t)parity * npart;
        const int64_t *pi_pre = a->part_i + (int64_t)parity * npart;
        void *kargs[] = {&sin_pre, &pf_pre, &pi_pre, &npart, &args, &plan, &parity, &mode, &xa};
        hipKernelNodeParams kp = {};
        kp.func = x ? de_p2p_kernel(a->fun_id, a->n, a->P, a->strategy, a->constraints)
                    : de_chain_kernel(a->fun_id, a->n, a->P, a->strategy, a->constraints);
        kp.gridDim = dim3(g.blocks + (x ? 1u : 0u));
        kp.blockDim = dim3(g.threads);
        kp.sharedMemBytes = (unsigned)g.lds;
        kp.kernelParams = kargs;
        kp.extra = nullptr;
        hipGraphNode_t node;
        SX_HIP(hipGraphAddKernelNode(&node, gr->graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        prev = node;
    }
    SX_HIP(hipGraphInstantiate(&gr->exec, gr->graph, nullptr, nullptr, 0));
    *out = gr;
    return 0;
}

extern "C" int sx_de_chain_launch(const sx_de_args *a, int parity, int finalize_only, void *stream) {
    return chain_launch(a, nullptr, parity, finalize_only, stream);
}
extern "C" int sx_de_chain_graph_create(const sx_de_args *a, int ngen, int start_parity, sx_graph **out) {
    return chain_graph_create(a, nullptr, ngen, start_parity, out);
}

// ---------------------------------------------------------------------------
// Multi-GPU, peer exchange: the chained kernel on this rank's shard (a->P rows from global row a->row0)
// ---------------------------------------------------------------------------
extern "C" int sx_de_p2p_launch(const sx_de_args *a, const sx_xchg_args *x, int parity, int finalize_only,
                                void *stream) {
    if (int rc = check_xchg_args(x)) return rc;
    return chain_launch(a, x, parity, finalize_only, stream);
}
extern "C" int sx_de_p2p_graph_create(const sx_de_args *a, const sx_xchg_args *x, int ngen, int start_parity,
                                      sx_graph **out) {
    if (int rc = check_xchg_args(x)) return rc;
    return chain_graph_create(a, x, ngen, start_parity, out);
}

extern "C" int sx_graph_launch(sx_graph *g, void *stream) {
    SX_REQUIRE(g && g->exec, "sx_graph_launch: null graph");
    SX_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return 0;
}

extern "C" int sx_graph_destroy(sx_graph *g) {
    if (!g) return 0;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->scratch) (void)hipFree(g->scratch);
    delete g;
    return 0;
}
