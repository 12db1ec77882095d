/*
 * stochopy_hip.h -- C ABI of the MI355X (gfx950) population-evaluation engine.
 *
 * Drop-in boundary for the per-generation hot path of keurfonluu/stochopy v2.3.0
 * (a pure-Python/numpy package: there is no FFI in the reference; these are the
 * entry points a `backend="hip"` branch of its backend hook would bind through
 * ctypes -- see INTEGRATION.md).  Each entry point cites the reference code it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types.
 *   - every `double*` / `int*` marked DEVICE points into HBM owned by the caller;
 *     the library never allocates or frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     All kernels are asynchronous on that stream.
 *   - return value: 0 = ok, negative = error (text via sx_last_error()).
 *   - float64 throughout, as in the reference; compiled
 *     with -ffp-contract=off so a*b+c rounds twice like numpy does.
 *   - populations are row-major (P, ld) with ld >= n (row stride in doubles).
 */
#ifndef STOCHOPY_HIP_H
#define STOCHOPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_ABI_VERSION 1

/* objective ids: stochopy/factory/benchmark.py:14-156 */
enum {
    SX_FUN_ACKLEY = 0,          /* benchmark.py:14-34   */
    SX_FUN_GRIEWANK = 1,        /* benchmark.py:37-56   */
    SX_FUN_QUARTIC = 2,         /* benchmark.py:59-76   */
    SX_FUN_RASTRIGIN = 3,       /* benchmark.py:79-97   */
    SX_FUN_ROSENBROCK = 4,      /* benchmark.py:100-118 */
    SX_FUN_SPHERE = 5,          /* benchmark.py:121-136 */
    SX_FUN_STYBLINSKI_TANG = 6, /* benchmark.py:139-156 */
    SX_FUN_COUNT = 7
};

/* DE strategies: stochopy/optimize/de/_strategy.py:1-46 */
enum { SX_DE_RAND1BIN = 0, SX_DE_RAND2BIN = 1, SX_DE_BEST1BIN = 2, SX_DE_BEST2BIN = 3 };

/* random-draw source for the generation kernels */
enum {
    SX_RNG_HOST = 0,   /* draws uploaded by the host (numpy-legacy MT19937 stream, parity mode) */
    SX_RNG_PHILOX = 1  /* Philox4x32-10 generated in-kernel, counter = (slot,row,gen,purpose)    */
};

/* termination status, stochopy/optimize/_common.py:13-24 */
#define SX_STATUS_NONE 100

int sx_abi_version(void);
const char *sx_last_error(void);
/* number of visible HIP devices (<0 on error); used by the loader to fail loudly */
int sx_device_count(void);
/* sizeof(sx_state / sx_de_args / sx_pso_args / sx_xchg_args) as compiled: lets a binding check its struct
 * mirror (which: 0 state, 1 DE args, 2 PSO args, 3 exchange args, 4 CMA state, 5 CMA args, 6 VD-CMA args; -1 otherwise) */
int sx_struct_size(int which);

/* ------------------------------------------------------------------------- *
 * Summation plan: numpy's pairwise add.reduce order for a length-m vector
 * (numpy/_core/src/umath/loops_utils.h.src, blocksize 128).
 * The objective kernels follow it so fitness values are bit-identical to the
 * reference's `.sum()` for +,-,* objectives.  Host function, exported so tests can
 * pin the order; the kernels receive the plan in their arguments (scalar loads),
 * built by the library from (fun_id, n).
 *   out[0]=nleaf out[1]=tail out[2]=full 8-blocks out[3]=stack depth
 *   out[4+2t]=end block of leaf t, out[5+2t]=merges after leaf t
 * Returns the number of int32 written (<= cap), or <0 if cap is too small.
 * ------------------------------------------------------------------------- */
int sx_sum_plan(int64_t m, int32_t *out, int cap);
/* number of terms the objective sums for an n-vector (n or n-1) */
int64_t sx_fun_terms(int fun_id, int n);

/* ------------------------------------------------------------------------- *
 * Batched objective evaluation  f[i] = fun(X[i,:])
 * replaces: the population wrapper stochopy/optimize/_common.py:34-90
 *           (serial :79-80, joblib :39-43, MPI :58-72) applied to the
 *           benchmark objectives, and cmaes/_cmaes.py:167-173 when xm/xstd are
 *           given (x -> x*xstd + xm before the objective).
 * X DEVICE (P,ldx); f DEVICE (P); xm/xstd DEVICE (n) or NULL.
 * part_f/part_i DEVICE (sx_num_partials(P,n)) or NULL: per-workgroup (min f, first row)
 * records for sx_select_finalize.  16, 32 or 64 lanes evaluate one individual; n <= 4096.
 * ------------------------------------------------------------------------- */
int64_t sx_num_partials(int64_t P, int n);
int sx_rows_per_workgroup(int n); /* rows of one workgroup of the row kernels = rows behind one (part_f, part_i) record */
/* Rows of more than this many elements are served by the one-workgroup-per-row kernels (one record per row, no chained /
 * peer-exchange form): what a host loop needs to know to size its record buffers and to pick the two-kernel path. */
int sx_wide_from(void);
/* The same threshold, set for the runs that follow: n <= 0 restores the library's own (2048: where the one-workgroup-per-row
 * kernels become the faster ones); otherwise clamped to [256, 4096], the range the wavefront-per-row kernels can serve.  A run
 * that needs the chained kernel's peer exchange or its global-donor gathers on rows of 2049 ... 4096 elements raises it to 4096
 * for its duration (optimize/_de.py).  Returns the previous value.  Process-wide, not thread-safe. */
int sx_set_wide_from(int n);
int sx_eval(int fun_id, const double *X, int64_t P, int n, int64_t ldx, const double *xm, const double *xstd,
            double *f, double *part_f, int64_t *part_i, void *stream);

/* Initial population with in-kernel draws (rng="philox"): the Latin hypercube of
 * _common.py:109-120 (strata of width 2/P, jitter of width 1/P, one stratum per row and column, scaled to the
 * bounds with the reference's two-rounding arithmetic), every row computable on its own: stratum = keyed bijection
 * of [0, P) per column (Philox keys), jitter = 53-bit Philox uniform keyed by (global row, element).  A rank of a
 * sharded run draws only its rows [row0, row0 + rows) of the P-row population.  X DEVICE (rows, ld); lower/upper
 * DEVICE (n).  Oracle counterpart: oracle/streams.py PhiloxStream.lhs_population. */
int sx_philox_lhs(double *X, int64_t rows, int n, int64_t ld, int64_t row0, int64_t P, const double *lower,
                  const double *upper, uint32_t key0, uint32_t key1, void *stream);

/* argmin with numpy's first-minimum tie rule (np.argmin, _common.py:132, de/_de.py:216)
 * f DEVICE (P); ws_f/ws_i DEVICE scratch of ws_len >= 1 entries; out_idx/out_val DEVICE (1). */
int sx_argmin(const double *f, int64_t P, double *ws_f, int64_t *ws_i, int64_t ws_len, int64_t *out_idx,
              double *out_val, void *stream);

/* ------------------------------------------------------------------------- *
 * Generation state shared by DE and PSO (64 bytes, DEVICE; host reads it back)
 * ------------------------------------------------------------------------- */
typedef struct sx_state {
    int64_t it;      /* generations completed (the reference's `it`; initial evaluation = 1) */
    int64_t gbidx;   /* row of the current best                                             */
    double gfit;     /* best fitness                                                         */
    double dx;       /* ||xbest_prev - xbest|| of the last selection (_common.py:135)        */
    int32_t status;  /* SX_STATUS_NONE while running, else -1 / 0 / 1 (_common.py:134-158)   */
    int32_t done;    /* 1 once status is set: later generation launches are no-ops           */
    int64_t reserved[3];
} sx_state;

/* ------------------------------------------------------------------------- *
 * Differential Evolution, synchronous generation
 * replaces: de/_de.py:314-351 de_sync (mutation via _strategy.py, binomial
 *           crossover `r1 <= CR` + forced index, de/_constraints.py:13-28 Random)
 *           + _common.py:123-160 selection_sync + the objective calls, fused.
 * Population storage: two row buffers buf0/buf1 (P,ld); generation g (= state->it)
 * lives in buf[g & 1].  A generation reads X_i and its k donor rows from the
 * current buffer and writes row i of the other one (the trial vector if it wins
 * with strict <, else the unchanged row), so donor reads never race with
 * selection writes and a generation moves (k+2) rows per individual
 * (SURVEY.md section 8d).
 * ------------------------------------------------------------------------- */
typedef struct sx_de_args {
    double *buf0, *buf1;    /* DEVICE (P,ld) each                                     */
    double *fit;            /* DEVICE (P) personal-best fitness (pbestfit)            */
    double *candfit;        /* DEVICE (P) candidate fitness `pfit` or NULL            */
    double *gbest;          /* DEVICE (n) copy of the best row (maintained by sx_select_finalize);
                               NULL = read row state->gbidx of the current buffer       */
    const double *lower;    /* DEVICE (n)                                             */
    const double *upper;    /* DEVICE (n)                                             */
    sx_state *state;        /* DEVICE                                                 */
    double *part_f;         /* DEVICE (sx_num_partials(P))                            */
    int64_t *part_i;        /* DEVICE (sx_num_partials(P))                            */
    /* SX_RNG_HOST inputs for ONE generation (the numpy-legacy stream, App. B):      */
    const double *r1;       /* DEVICE (P,n) rand(P,n), de/_de.py:250                  */
    const int32_t *donors;  /* DEVICE (k,P) first k rows of delete_shuffle_sync, :304 */
    const int32_t *irand;   /* DEVICE (P) randint(n,size=P), :340                     */
    const double *resample; /* DEVICE (P,n) uniform(lo,hi,(P,n)) or NULL, _constraints.py:24 */
    int64_t P;
    int64_t ld;
    int64_t row0;           /* global index of local row 0 (Philox counters; multi-GPU shards) */
    int32_t n;
    int32_t fun_id;
    int32_t strategy;
    int32_t constraints;    /* 0 none, 1 Random                                       */
    int32_t rng;            /* SX_RNG_HOST / SX_RNG_PHILOX                            */
    int32_t maxiter;
    double F, CR, xtol, ftol;
    uint32_t key0, key1;    /* Philox key = seed                                      */
} sx_de_args;

/* one generation: propose+evaluate+select kernel, then (if finalize) the
 * best/termination kernel.  finalize=0 is for multi-GPU, where the caller
 * exchanges the shard bests first and then calls sx_select_finalize itself. */
int sx_de_generation(const sx_de_args *a, int finalize, void *stream);

/* ------------------------------------------------------------------------- *
 * Best-of-generation + termination: _common.py:131-158 (argmin, xtol/ftol/maxiter ladder)
 * Reduces the per-workgroup partials, computes dx against `gbest`, sets status,
 * copies the new best row into `gbest`, increments state->it.
 * The best row is read from rows[(state->it+1) & 1] (the generation being
 * finalised; pass rows0 == rows1 for state that is updated in place, e.g. PSO pbest).
 * Rows of more than 4096 elements (three launches: the best record, the row's slices on up to 64 workgroups, the state):
 * the records are CONSUMED -- the first min(npart, 64) doubles of part_f are reused for the slices' partial squared
 * distances, so part_f holds no records afterwards (it is declared const for the common case; read the records, if
 * wanted, before this call).
 * ------------------------------------------------------------------------- */
int sx_select_finalize(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                       const double *rows1, int64_t ld, int n, double *gbest, sx_state *state, int maxiter,
                       double xtol, double ftol, void *stream);

/* ------------------------------------------------------------------------- *
 * Multi-GPU (population sharded by rows, one process per GPU): the per-generation exchange that
 * takes the place of the reference's MPI Bcast/Allreduce (stochopy/optimize/_common.py:58-72).
 * sx_shard_best: this shard's best of the generation being finalised ->
 *     record (n+2 doubles) = [ f, (double)(row0 + local row), row[0..n) ]
 * The caller all-gathers the records of all ranks (RCCL over xGMI) into `records` (world,(n+2)).
 * sx_gather_finalize: first minimum over the records by (f, global row) = np.argmin over the whole
 *     population, then the same dx / status / gbest / state update as sx_select_finalize.
 * ------------------------------------------------------------------------- */
int sx_shard_best(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                  const double *rows1, int64_t ld, int n, const sx_state *state, int64_t row0, double *record,
                  void *stream);
int sx_gather_finalize(const double *records, int world, int n, double *gbest, sx_state *state, int maxiter,
                       double xtol, double ftol, void *stream);
/* sx_de_generation(a, 0) + sx_shard_best in one host call (fewer launches' worth of host time per generation) */
int sx_de_shard_generation(const struct sx_de_args *a, double *record, void *stream);

/* ------------------------------------------------------------------------- *
 * hipGraph of `ngen` identical generations (all per-generation state lives in
 * `state` on the device, so one instantiated graph is replayed).
 * Only valid for SX_RNG_PHILOX (host draws change every generation).
 * ------------------------------------------------------------------------- */
typedef struct sx_graph sx_graph;
int sx_de_graph_create(const sx_de_args *a, int ngen, sx_graph **out);
/* Chained finalize (single GPU, SX_RNG_PHILOX): ONE kernel per generation.  a->state must point to
 * sx_state[3] and a->part_f / a->part_i to [2][sx_num_partials(P,n)].  Launch L uses parity L & 1: every
 * workgroup first re-reduces the records its predecessor wrote to part[parity] (the best-of-generation /
 * termination step of _common.py:131-158, published in state[1-parity] by workgroup 0) and then produces
 * the next generation with records in part[1-parity].  finalize_only: just that first half, result in
 * state[2] (what the host reads).  The kernel stops on fun <= ftol with status 1; status 0 (xtol) is settled
 * by the caller from state.reserved[0] = the previous best row.  Start: state[0] = {it 0, gbidx}, part[0] =
 * {(f_best, row_best), +inf...}, generation 1 in buf1. */
int sx_de_chain_launch(const sx_de_args *a, int parity, int finalize_only, void *stream);
int sx_de_chain_graph_create(const sx_de_args *a, int ngen, int start_parity, sx_graph **out);
int sx_graph_launch(sx_graph *g, void *stream);
int sx_graph_destroy(sx_graph *g);

/* ------------------------------------------------------------------------- *
 * Multi-GPU, peer exchange over xGMI (one process per GPU; population sharded by rows)
 * replaces: the per-generation MPI traffic of stochopy/optimize/_common.py:58-72 (Bcast + Allreduce)
 *           for the global best of _common.py:131-133.
 * Every rank owns an exchange buffer in its HBM (uncached, IPC-shareable) that all peers map.  It holds
 * slots[2][world][2*(n+2)] 64-bit words: the record [f, global row, best row] of each rank, each 32-bit
 * half carried with a 32-bit generation tag in one 8-byte store (arrival is detected on the data itself,
 * no separate flag, no fence).  sx_de_p2p_* = the chained kernel of sx_de_chain_* where, per generation,
 * workgroup 0 writes this shard's record straight into every peer's slot and every wavefront picks the
 * global best out of its own rank's slots as they arrive: still ONE kernel per generation and no
 * collective call on the data path.  A rank that waits longer than `timeout_ticks` (100 MHz ticks) sets
 * *error and the kernels become no-ops (the caller raises).
 * Start as for sx_de_chain_launch with a->gbest = NULL; afterwards the best row of the last generation is
 * slot[parity][state.reserved[1]] (decode with sx_xchg_read_record), the previous one is in the other parity.
 * ------------------------------------------------------------------------- */
#define SX_MAX_PEERS 8
#define SX_IPC_HANDLE_BYTES 64
typedef struct sx_xchg_args {
    uint64_t *peer[SX_MAX_PEERS]; /* DEVICE: exchange buffer of rank r as mapped in this process (peer[rank] = own) */
    int32_t world, rank;
    int64_t timeout_ticks;        /* per wait, in 100 MHz ticks (1e8 = 1 s) */
    int32_t *error;               /* DEVICE: 0, set to 1 on timeout */
    uint64_t *relay;              /* DEVICE, ordinary (cacheable) memory, sx_xchg_relay_bytes(n), zeroed: rows of
                                     more than 128 elements are read from here after workgroup 0 has copied the
                                     winning record out of the uncached exchange buffer (once per generation,
                                     instead of once per wavefront) */
    /* donors over the WHOLE population (exact reference semantics, de/_de.py:304-311, at the price of remote
     * row reads over xGMI): global_rows = world * shard_rows > 0 and pop0/pop1[r] = rank r's two population
     * buffers (sx_pop_alloc + sx_xchg_open).  global_rows = 0: donors are drawn inside the shard. */
    const double *pop0[SX_MAX_PEERS];
    const double *pop1[SX_MAX_PEERS];
    int64_t global_rows, shard_rows;
} sx_xchg_args;
/* bytes of one exchange buffer for `world` ranks and rows of n doubles (includes the probe area) */
int64_t sx_xchg_bytes(int world, int n);
int64_t sx_xchg_relay_bytes(int n);
/* allocate + zero an exchange buffer on the current device and export it (handle: 64 bytes) */
int sx_xchg_alloc(int64_t bytes, void **ptr, void *handle);
int sx_xchg_free(void *ptr);
/* ordinary device memory that peers can map (population buffers read remotely with global donors); release with
 * sx_xchg_free, map / unmap with sx_xchg_open / sx_xchg_close */
int sx_pop_alloc(int64_t bytes, void **ptr, void *handle);
/* map a peer's buffer from the handle it exported / unmap it */
int sx_xchg_open(const void *handle, void **ptr);
int sx_xchg_close(void *ptr);
/* transport self-test: `rounds` tagged all-to-all writes + waits through the probe area of the buffers;
 * returns 0 = every word of every round arrived, 1 = timeout or corrupt word (synchronises the stream) */
int sx_xchg_probe(const sx_xchg_args *x, int n, int rounds, void *stream);
/* decode slot[parity][src] of the own buffer into record[n+2] (host memory); synchronises the stream */
int sx_xchg_read_record(const sx_xchg_args *x, int n, int parity, int src, double *record, void *stream);
/* The same exchange as ONE one-workgroup kernel for generation kernels that are not chained (PSO / CPSO):
 * sx_shard_best + all-gather + sx_gather_finalize without a collective -- shard best from the workgroup
 * records, record into every peer's slot, wait for all ranks, global best, dx, gbest, status, it++
 * (_common.py:131-158).  rows0/rows1, ld, row0 as for sx_shard_best; gbest/state as for sx_gather_finalize. */
int sx_xchg_finalize(const double *part_f, const int64_t *part_i, int64_t npart, const double *rows0,
                     const double *rows1, int64_t ld, int n, int64_t row0, double *gbest, sx_state *state, int maxiter,
                     double xtol, double ftol, const sx_xchg_args *x, void *stream);
int sx_de_p2p_launch(const sx_de_args *a, const sx_xchg_args *x, int parity, int finalize_only, void *stream);
int sx_de_p2p_graph_create(const sx_de_args *a, const sx_xchg_args *x, int ngen, int start_parity, sx_graph **out);

/* ------------------------------------------------------------------------- *
 * PSO / CPSO, synchronous generation
 * replaces: cpso/_cpso.py:324-329 mutation (V = w*V + c1*r1*(pbest-X) + c2*r2*(gbest-X),
 *           left-to-right), cpso/_constraints.py:4-10 / 44-53 (X+V, or Shrink: per-row
 *           beta = min over violated dims of (bound-x)/v), cpso/_cpso.py:332-361 pso_sync
 *           + _common.py:123-160 selection_sync (cand = X, x = pbest) + the objective.
 * X, V, pbest, pbestfit are updated in place (row-local state).
 * ------------------------------------------------------------------------- */
typedef struct sx_pso_args {
    double *X, *V, *pbest;  /* DEVICE (P,ld) each                                     */
    double *pbestfit;       /* DEVICE (P)                                             */
    double *candfit;        /* DEVICE (P) fitness of the new positions `pfit`, or NULL */
    double *gbest;          /* DEVICE (n) copy of the best row (sx_select_finalize)   */
    const double *lower;    /* DEVICE (n)                                             */
    const double *upper;    /* DEVICE (n)                                             */
    sx_state *state;        /* DEVICE                                                 */
    double *part_f;         /* DEVICE (sx_num_partials(P,n))                          */
    int64_t *part_i;        /* DEVICE (sx_num_partials(P,n))                          */
    const double *r1;       /* DEVICE (P,n) rand(P,n), cpso/_cpso.py:262  (SX_RNG_HOST) */
    const double *r2;       /* DEVICE (P,n) rand(P,n), cpso/_cpso.py:263  (SX_RNG_HOST) */
    const uint64_t *pending_restart; /* DEVICE (3) or NULL: an sx_pso_restart_select decision (its out3) of the
                             * PREVIOUS generation that sx_pso_restart_apply has not carried out: the generation kernel
                             * re-seeds those rows itself (same Philox positions, V = 0, pbest = X, pbestfit = 1e30,
                             * cpso/_cpso.py:420-424) instead of loading them.  Set inside sx_pso_graph_create only
                             * (in-kernel draws); callers pass NULL and apply restarts with sx_pso_restart_apply. */
    int64_t P;
    int64_t ld;
    int64_t row0;           /* global index of local row 0 (Philox counters; shards)  */
    int32_t n;
    int32_t fun_id;
    int32_t constraints;    /* 0 none, 1 Shrink                                       */
    int32_t rng;
    int32_t maxiter;
    int32_t pad_;
    double w, c1, c2, xtol, ftol;
    uint32_t key0, key1;
} sx_pso_args;

int sx_pso_generation(const sx_pso_args *a, int finalize, void *stream);

/* PSO with ONE kernel per generation (round 3; the pattern of sx_de_chain_launch): the best / termination step of
 * _common.py:131-158 for generation g runs in the prologue of the launch that produces generation g+1 -- every
 * workgroup reduces the per-workgroup records (two levels) and derives the same best row and status.  Because the
 * swarm is updated in place, each workgroup keeps a copy of its best row in best_rows (double-buffered; the record
 * says which copy is current), which is where the next generation reads gbest from.
 * Supported (sx_pso_chain_supported != 0): one GPU, SX_RNG_PHILOX, constraints none, no pending restart, whole-batch
 * rows (n = 64, 128 or 256), sx_num_partials(P,n) <= 8 x the workgroup size.
 * a->state: 3 sx_state words ([0], [1] ping-pong by launch parity, [2] written by finalize_only launches = the
 * host's view; reserved[0] / reserved[1] = record of the previous / current best); a->part_f / a->part_i:
 * 2 x npart records, record = 2 * row + q; best_rows DEVICE (2, npart, n); a->gbest unused.  status 1 is reported
 * for `fun <= ftol`; whether it is 0 (best moved by <= xtol) the host settles from the two resident rows. */
int sx_pso_chain_supported(const sx_pso_args *a);
int sx_pso_chain_launch(const sx_pso_args *a, double *best_rows, int parity, int finalize_only, void *stream);
int sx_pso_chain_graph_create(const sx_pso_args *a, double *best_rows, int ngen, int start_parity, sx_graph **out);

/* Competitive restart, cpso/_cpso.py:405-426.
 * sx_pso_radius: part_r[b] = max over the rows of workgroup b of ||X_i - gbest||_2 (:410)
 *   part_r DEVICE (sx_num_partials(P,n)).
 * sx_pso_restart_select (one workgroup; keys in registers up to 32768 particles, re-read from L2 above): radius = max(part_r)/sqrt(4n); if
 *   radius < delta: nw = int((P-1)/(1+exp((it/maxiter-gamma+0.5)/0.09))) and the nw-th largest
 *   pbestfit (radix descent).  out3 DEVICE uint64[3] = {nw, threshold key, radius bits}.
 * sx_pso_restart_apply: rows with pbestfit among the nw worst get V=0, X=uniform(lower,upper),
 *   pbest=X, pbestfit=1e30 (:420-424).  Either sel3 (device selection, Philox positions keyed
 *   by row) or host_rows/host_x (DEVICE copies of the reference-ordered row ids and their
 *   numpy-legacy positions, host_count rows). */
int sx_pso_radius(const sx_pso_args *a, double *part_r, void *stream);
int sx_pso_restart_select(const sx_pso_args *a, const double *part_r, double delta, double gamma, uint64_t *out3,
                          void *stream);
int sx_pso_restart_apply(const sx_pso_args *a, const uint64_t *sel3, const int64_t *host_rows, const double *host_x,
                         int64_t host_count, void *stream);
/* Sharded swarm (one process per GPU, a->P rows each): the same selection over the WHOLE swarm.
 * gathered DEVICE (world, a->P + sx_num_partials(a->P, n)): row r = rank r's [pbestfit | part_r] after one
 * all-gather per generation (the allreduce(max) of the radius and the fitness all-gather of SURVEY.md
 * section 8e in one message).  Every rank computes the same {nw, threshold, radius}. */
int sx_pso_restart_select_gathered(const sx_pso_args *a, const double *gathered, int world, double delta,
                                   double gamma, uint64_t *out3, void *stream);
/* hipGraph of `ngen` generations of the cpso loop body (cpso/_cpso.py:257-307), single GPU + SX_RNG_PHILOX:
 * per generation sx_pso_generation(a, 1) and, when part_r/sel3 are given (CPSO), the radius and selection kernels; a
 * decided restart is carried out by the next generation's kernel (sx_pso_args.pending_restart, set here), the apply
 * kernel runs once after the last generation, so X / V / pbest are complete whenever a replay has finished.
 * Replay with sx_graph_launch; launches after convergence are no-ops. */
int sx_pso_graph_create(const sx_pso_args *a, int ngen, double *part_r, double delta, double gamma, uint64_t *sel3,
                        sx_graph **out);

/* ------------------------------------------------------------------------- *
 * Generations around a CALLER-SUPPLIED objective (the backend hook contract of _common.py:27-106: `fun(X)` maps
 * the (P,n) population to (P,) values -- here a callable working on the device array).  The objective cannot
 * be fused, so a generation is: sx_de_propose / sx_pso_move -> caller's objective -> sx_rows_select ->
 * sx_select_finalize (or the shard-best exchange).  Same draws / arithmetic / records as the fused kernels.
 *   sx_de_propose : cand (P,n) row-major <- the trial vectors of generation state->it + 1 (de/_de.py:333-344,
 *                   de/_strategy.py, de/_constraints.py:13-28); nothing else is written.
 *   sx_pso_move   : V, X updated in place (cpso/_cpso.py:324-329, cpso/_constraints.py:4-53).
 *   sx_rows_select: selection_sync after the evaluation (_common.py:127-129): rows with f[i] < xfun[i] take
 *                   cand[i] (xout row <- cand row, xfun[i] <- f[i]); the others keep xin's row (copied when
 *                   xout != xin: DE's other population buffer); candfit[i] <- f[i] (or NULL); part_f/part_i <-
 *                   the workgroup records sx_select_finalize / sx_shard_best expect. */
int sx_de_propose(const sx_de_args *a, double *cand, void *stream);
int sx_pso_move(const sx_pso_args *a, void *stream);
int sx_rows_select(const double *cand, int64_t ldc, const double *f, const double *xin, double *xout, int64_t ldx,
                   double *xfun, double *candfit, int64_t P, int n, const sx_state *state, double *part_f,
                   int64_t *part_i, void *stream);

/* ------------------------------------------------------------------------- *
 * updating="immediate": ONE asynchronous generation, i.e. the ordered sweep in which individual i sees
 * what individuals 0..i-1 did in the same generation (csrc/sx_async.hip: one workgroup; rounds of up to 64
 * individuals are proposed together, judged in order, and the few that depended on an earlier one of their
 * round are proposed again -- the sequential result, bit for bit).
 * replaces: de/_de.py:354-391 de_async, cpso/_cpso.py:364-402 pso_async, _common.py:163-194 selection_async
 *           (`<=` acceptance, best row + status updated per individual, the last individual's status wins,
 *           then `it >= maxiter -> -1`), cpso/_constraints.py:56-64 (Shrink, one-row form), objective fused.
 * State: population in place (DE: a->buf0 only; PSO: X, V, pbest), a->gbest (n) in/out = the best row,
 * a->state->it / gfit in/out, status / done out.  SX_RNG_HOST inputs as for the synchronous generation, but
 * drawn in the asynchronous order (sx_mt_de_async_draws).  part_f / part_i / buf1 are not used.
 * Single GPU (row0 = 0, P = whole population). */
int sx_de_async_generation(const sx_de_args *a, void *stream);
int sx_pso_async_generation(const sx_pso_args *a, void *stream);

/* ------------------------------------------------------------------------- *
 * CMA-ES device kernels (fp64 MFMA, v_mfma_f64_16x16x4_f64, LDS-tiled 64x64x32)
 * sx_cmaes_sample   replaces cmaes/_cmaes.py:232-237:
 *     arx[i,:] = xmean + sigma * dot(B, D * Z[i,:])          Z (P,n) standard normals
 * sx_cmaes_recombine replaces cmaes/_cmaes.py:274:
 *     xmean = dot(w, arx[idx[:mu], :])                        idx DEVICE int64 (mu), best first
 * sx_cmaes_rank_mu  replaces cmaes/_cmaes.py:290-295 (in place on C, full matrix as the reference):
 *     artmp = (arx[idx[:mu]] - xold) / sigma
 *     C = C*(1-c1-cmu) + cmu * artmp^T diag(w) artmp + c1 * outer(pc,pc) + tmp_coef * C_old
 *     with tmp_coef = 0 when `cond` holds, else c1*cc*(2-cc)  (:291)
 * sx_cmaes_normals  in-kernel Philox Box-Muller normals (throughput mode of :234)
 * sx_symmetrize_upper replaces cmaes/_cmaes.py:303: C = triu(C) + triu(C,1).T
 * All pointers DEVICE.  The eigendecomposition (:304, numpy/LAPACK in the reference) and the
 * scalar path/step-size/convergence logic stay on the host.
 * ------------------------------------------------------------------------- */
int sx_cmaes_sample(const double *xmean, double sigma, const double *B, const double *D, const double *Z, double *arx,
                    int64_t P, int n, void *stream);
int sx_cmaes_recombine(const double *arx, const int64_t *idx, const double *w, int mu, int n, double *xmean,
                       void *stream);
int sx_cmaes_rank_mu(const double *arx, const int64_t *idx, const double *w, int mu, const double *xold, double sigma,
                     const double *pc, double c1, double cmu, double tmp_coef, double *C, double *ws_y, int n,
                     void *stream); /* ws_y: DEVICE scratch (mu,n) for artmp */
int sx_cmaes_normals(double *Z, int64_t P, int n, int64_t row0, uint32_t gen, uint32_t key0, uint32_t key1,
                     void *stream);
int sx_symmetrize_upper(double *C, int n, void *stream);
/* CMA-ES box-constraint handling "Penalize", device part.
 * replaces: cmaes/_constraints.py:29-31 (candidates clipped to the standardised box [-1,1]^n, then the objective)
 *           and :79 (arfitness += dot((arxvalid - arx)**2, bnd_weights / bnd_scale)); the scalar bookkeeping
 *           of :33-76 (percentiles, weight history) stays on the host as in the reference.
 * X DEVICE (P,n) standardised candidates; xm/xstd DEVICE (n); v DEVICE (n) = bnd_weights / bnd_scale or NULL;
 * f_raw DEVICE (P) = fun(unstandardise(clip(X))); pen DEVICE (P) = sum_j (clip(x_ij) - x_ij)^2 v_j (NULL with v). */
int sx_cmaes_eval_penalized(int fun_id, const double *X, int64_t P, int n, const double *xm, const double *xstd,
                            const double *v, double *f_raw, double *pen, void *stream);
/* VD-CMA sampling, vdcma/_vdcma.py:236-248: for candidates row0 .. row0+P-1 of the generation
 *   ary[i] = dvec o (Z[i] + coef * (Z[i] . vn) * vn)   (coef = sqrt(1 + |v|^2) - 1),   arx[i] = xmean + sigma * ary[i];
 * dy != NULL: global rows 0 and 1 become +dy / -dy (mean-shift injection, :241-247).
 * Z, ary, arx DEVICE (P,n); dvec, vn, xmean, dy DEVICE (n).  One wavefront per candidate, O(n). */
int sx_vdcma_sample(const double *Z, int64_t P, int n, int64_t row0, const double *dvec, const double *vn, double coef,
                    const double *xmean, double sigma, const double *dy, double *ary, double *arx, void *stream);

/* ------------------------------------------------------------------------- *
 * CMA-ES, device-resident generation (csrc/sx_cma_loop.hip): one call enqueues a whole generation --
 * normals, sampling (MFMA), objective, ranking, recombination, evolution paths, step size, covariance
 * update (MFMA), [symmetrise + sx_eigh], the ten stopping rules -- and the host looks at the 128-byte
 * state only every few generations.
 * replaces cmaes/_cmaes.py:226-343 (one pass of the `while` loop) and :360-434 (`converge`), for draws made on the
 * device (Philox), constraints=None, one GPU.  Everything in sx_cma_args is DEVICE memory unless stated.
 *   gen       1-based generation number (the caller counts; it also keys the Philox normals)
 *   do_eigh   the caller's evaluation of :301 (`nfev - eigeneval > popsize / (c1 + cmu) / ndim / 10`, a function of
 *             gen only): 0 not due, 1 decompose, 2 decompose starting from the previous eigenvectors in B
 * After a stopping rule has fired (state->done) the bookkeeping kernels of later calls do nothing and
 * xbest / state keep the result; besthist must be zero-initialised (the reference's np.zeros(maxiter), :220).
 * ------------------------------------------------------------------------- */
typedef struct sx_cma_state {
    int64_t it;         /* generations completed                                                       */
    int64_t nfev;       /* it * popsize                                                                */
    int64_t best_row;   /* arindex[0] of the last generation                                           */
    double fbest;       /* arfitness[arindex[0]]                                                       */
    double sigma;       /* step size entering the next generation                                      */
    double sigma_next;  /* scratch: step size after :298, published by the stop step                   */
    double tmp_coef;    /* 0 when `cond` held (:283-291), else c1*cc*(2-cc)                            */
    double psnorm;      /* |ps|                                                                        */
    int32_t status;     /* SX_STATUS_NONE while running, else the reference's status (-8 .. 1)         */
    int32_t done;       /* 1 once a stopping rule fired                                                */
    int64_t stop_it;    /* generation at which it fired                                                */
    double reserved[6];
} sx_cma_state;

typedef struct sx_cma_args {
    double *Z, *arx;            /* (P,n) normals / candidates (standardised coordinates)              */
    double *fit;                /* (P)                                                                */
    double *xmean, *xold, *ps, *pc; /* (n)                                                            */
    double *C, *B;              /* (n,n) covariance, eigenvectors in columns                          */
    double *D;                  /* (n) sqrt of the eigenvalues                                        */
    double *eigw;               /* (n) eigenvalues as sx_eigh returns them                            */
    const double *w;            /* (mu) recombination weights                                         */
    double *Y;                  /* (mu,n) scratch of the covariance update                            */
    double *part;               /* (64,n) scratch of the recombination                                */
    double *step, *isc, *xnew;  /* (n) scratch: xmean - xold, C^(-1/2) step, the new mean             */
    double *ypart;              /* (8,n) scratch of B^T step                                          */
    double *besthist;           /* (maxiter) zero-initialised                                         */
    const double *xm, *xstd;    /* (n) un-standardisation x * xstd + xm (:167-173)                    */
    double *xbest;              /* (n) result: best candidate of the stopping generation              */
    double *hist_x, *hist_f;    /* return_all history (maxiter, rows, n) / (maxiter, rows), un-standardised candidates
                                 * (:262-269), or NULL; rows = hist_rows, or 1 when hist_rows == 0 (best candidate only) */
    int64_t *order;             /* (P) argsort of fit                                                 */
    void *state;                /* sx_cma_state                                                       */
    void *eigh_ws;              /* sx_eigh workspace                                                  */
    double *pen_ws;             /* constraints="Penalize" on the device (cmaes/_constraints.py:4-82), or NULL: doubles
                                 * [weights n | v n | penalty P | spread history 256 | count, validfitval, iniphase, -];
                                 * initialise weights = 0, history[0] = 1, count = 1, validfitval = 0, iniphase = 1.  Needs
                                 * int(20 + 3n/P) + 1 <= 256 history entries (else: the host-driven loop)   */
    int64_t *pen_order;         /* (P) argsort of the raw fitness (percentiles of :34-35), with pen_ws      */
    int64_t eigh_ws_bytes;
    int64_t P;
    int64_t hist_rows;          /* ceil(verbosity * popsize)                                          */
    int32_t n, mu, fun_id, maxiter, ilim, eig_sweeps;
    double cs, cc, c1, cmu, damps, chind, mueff, xtol, ftol, insigma;
    uint32_t key0, key1;
} sx_cma_args;

int sx_cmaes_generation(const sx_cma_args *a, int64_t gen, int do_eigh, void *stream);
/* The same generation in two steps for candidates sharded over ranks (workers > 1; what the reference's parallel backends
 * shard: _common.py:58-72).  stage 0: this rank's candidates [row0, row0 + rows) -- Philox normals keyed by the global
 * row, sampling GEMM, objective -- into arx_loc (rows,n) / fit_loc (rows); the caller
 * all-gathers them into a->arx / a->fit; stage 1: everything else (ranking ... stop rules; Penalize's bookkeeping and
 * penalty pass), replicated on every rank.  stage 0 with all rows + stage 1 == sx_cmaes_generation.
 * a->Z must be the struct's full (P, n) buffer on EVERY rank, not a per-shard one: stage 0 uses its first rows x n doubles
 * for the normals, and stage 1's covariance update takes 2 n n doubles of it as split-K scratch whenever P >= 2 n (always
 * inside the (P, n) buffer then; with P < 2 n the update runs unsplit and needs none). */
int sx_cmaes_generation_stage(const sx_cma_args *a, int64_t gen, int do_eigh, int stage, int64_t row0, int64_t rows,
                              double *arx_loc, double *fit_loc, void *stream);
/* The generation with its decomposition (cmaes/_cmaes.py:301-309; do_eigh != 0) enqueued in pieces, one GPU: phase 0 =
 * candidates + model update up to and including the decomposition's start and its rounds [0, r1); the caller reads the
 * eigensolver's run record (the head of a->eigh_ws: int32 done_seq, sweeps, parity, converged; sx_eigh_info) and either adds
 * rounds (phase 1: [r0, r1)) or lets phase 2 (r1 = rounds enqueued so far) finish the decomposition and apply the stop rules.
 * Same kernels in the same order as sx_cmaes_generation, minus the no-op launches behind the run's end (a sweep's worth per
 * decomposition when the allowance has to be guessed in advance).  sx_eigh_rounds_per_sweep(n): rounds of one sweep of the
 * block method; 0 for the one-workgroup solver of small n, which cannot be enqueued in pieces. */
int sx_eigh_rounds_per_sweep(int n);
int sx_cmaes_generation_phased(const sx_cma_args *a, int64_t gen, int do_eigh, int phase, int r0, int r1, void *stream);

/* ------------------------------------------------------------------------- *
 * Symmetric eigendecomposition on the device (csrc/sx_eigh.hip): parallel two-sided block Jacobi, the
 * similarity updates as fp64 MFMA tile products, convergence decided on the device.
 * replaces cmaes/_cmaes.py:303-305:
 *     C = triu(C) + triu(C,1).T;  D, B = np.linalg.eigh(C);  idx = argsort(D);  D = D[idx];  B = B[:, idx]
 * (numpy.linalg.eigh = LAPACK dsyevd, the reference's third-party call).
 * C DEVICE (n,n) row-major, only its upper triangle is read (mirrored, as :303 does); w DEVICE (n) eigenvalues
 * ascending; B DEVICE (n,n) row-major, eigenvector k in column k, unit norm, CANONICAL SIGN: the component of
 * largest magnitude (lowest row on ties) is positive.  V0: NULL, or DEVICE (n,n) nearly orthonormal starting basis
 * (e.g. the eigenvectors of the previous, slightly different matrix; may alias B): it is re-orthonormalised (one
 * Newton-Schulz step) and the iteration starts from V0^T C V0 -- same result to rounding, fewer sweeps; ignored for
 * n <= 32 (one-workgroup path).  ws: DEVICE scratch of
 * sx_eigh_workspace_bytes(n) bytes; it starts with the run record read by sx_eigh_info.  max_sweeps <= 0: 24;
 * tol <= 0: 1e-14 (a sweep is the last one when the off-diagonal mass it leaves behind -- measured on the device
 * while its last rotations are applied -- is <= tol*|C|_F).
 * Asynchronous on `stream`; the host never waits: launches after convergence are no-ops.
 * sx_eigh_info (synchronises): sweeps carried out, whether the rule was met, off-diagonal mass / |C|_F left
 * behind by the last sweep.
 * sx_eigh_set_refine(mode): the last sweep of a run may be replaced by the first-order refinement step
 * V <- V (I + K + K K / 2), K_ij = M_ij / (M_jj - M_ii), once what the sweeps left is small against every gap
 * (off(M) <= 1e-7 |C|_F and max |K_ij| <= 1e-3, both measured on the device; csrc/sx_eigh.hip kRefineOff).
 * mode 1: always allowed; 0: never; -1 (initial): allowed inside the CMA-ES generation loops (sx_cmaes_generation*),
 * not in sx_eigh itself; the environment variable SX_EIGH_REFINE = 0 / 1 sets the initial mode.  Returns the
 * previous mode.  Process-wide, not thread-safe.
 * sx_eigh_set_flow(mode): how the rounds of a run are enqueued (n > 32).  0 (the default): one launch per round.
 * 1: ONE resident launch works through all rounds -- pair workgroups hand their rotations to their two successors through
 * agent-scope words, tile workgroups follow behind counters; every wait is bounded (SX_EIGH_FLOW_TIMEOUT_MS, default 2000: a
 * run whose wait ran out is reported like one that did not converge).  Identical results, bit for bit; measured on MI355X it
 * is no faster (profiles/r6_eigh_flow.txt), hence not the default.  The resident form needs its whole grid (at most one
 * workgroup per CU) on the chip at once: PROCESSES THAT SHARE ONE GPU MUST NOT USE IT.  -1: back to the initial mode
 * (environment variable SX_EIGH_FLOW = 0 / 1, else 0).  Returns the previous mode; -2 changes nothing and returns the mode in
 * effect (0 / 1).  Process-wide, not thread-safe.
 * ------------------------------------------------------------------------- */
int64_t sx_eigh_workspace_bytes(int n);
int sx_eigh(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes, int max_sweeps,
            double tol, void *stream);
/* sx_eigh with the refinement step allowed (refine > 0) / forbidden (0) for THIS call only (< 0: the process-wide mode):
 * what a multi-threaded host uses instead of flipping sx_eigh_set_refine around a call. */
int sx_eigh_refined(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes,
                    int max_sweeps, double tol, int refine, void *stream);
int sx_eigh_info(const void *ws, int *sweeps, int *converged, double *off_rel, void *stream);
int sx_eigh_set_refine(int mode);
int sx_eigh_set_flow(int mode);

/* VD-CMA: everything of the model update that is O(mu n), on the device.
 * replaces vdcma/_vdcma.py:289-295 (w . arx[arindex[:mu]]), :317 (w . ary[arindex[:mu]]) and :331-339 with :428-444 (the
 * weighted moments p, q of the selected steps under D (I + v v^T) D):
 *   out[0*n ..] = sum_k w_k arx[idx_k]        out[1*n ..] = sum_k w_k ary[idx_k]
 *   out[2*n ..] = p_mu                        out[3*n ..] = q_mu                 (y_k = ary[idx_k] / dvec, t_k = y_k . vn)
 * arx, ary DEVICE (P,n); idx DEVICE int64 (mu) best first; w, dvec, vn DEVICE; ws DEVICE scratch of
 * (mu rounded up to 8) + 4*64*n doubles; out DEVICE (4,n).  The host keeps the O(n) natural-gradient step. */
int sx_vdcma_moments(const double *arx, const double *ary, const int64_t *idx, const double *w, int mu, int n,
                     const double *dvec, const double *vn, double norm_v2, double *ws, double *out, void *stream);

/* VD-CMA, device-resident generation (csrc/sx_cma_loop.hip): the whole loop body of vdcma/_vdcma.py:232-425 between two
 * looks of the host at the state -- normals, the mean-shift injection (:241-247), candidates, objective, ranking, the
 * O(mu n) moment sums, and in ONE workgroup the O(n) part: mean / step (:292-295), the rank-gap step size (:298-306),
 * the evolution path (:309-314), alpha / beta (:317-328), the moments of the path, the natural gradient and the update
 * of v and d (:331-378), the stopping rules (cmaes/_cmaes.py:360-434 without the rules that need B, D).
 * state: an sx_cma_state whose reserved[0..4] = {ps, |v|^2, |v|, injection flag, sqrt(1 + |v|^2) - 1}; sigma_next,
 * tmp_coef, psnorm unused.  All vectors DEVICE; besthist zero-initialised; hist_x / hist_f as in sx_cma_args or NULL. */
typedef struct sx_vd_args {
    double *Z, *ary, *arx;      /* (P,n) normals / steps y / candidates x.  Wide models (n > 4096) without hist_x and pen_ws:
                                 * arx may be NULL -- x = xmean + sigma y is then not kept (a quarter of a generation's memory
                                 * traffic): the moment sums and the result form it again from y, same bits.  Wide models use
                                 * Z for t_k (P doubles) and eight words behind them only.                                    */
    double *fit;                /* (P)                                                                */
    double *xmean, *xold, *dx, *dvec, *vvec, *vn, *pc; /* (n)                                         */
    double *zinj, *dy;          /* (n) the injection's normal row ("row P" of the generation) and +-dy */
    const double *w;            /* (mu)                                                               */
    double *mws, *mout;         /* sx_vdcma_moments' workspace and its (4,n) output                    */
    double *besthist;           /* (maxiter)                                                          */
    const double *xm, *xstd;    /* (n) un-standardisation                                             */
    double *xbest;              /* (n) result                                                         */
    double *hist_x, *hist_f;    /* return_all history slabs or NULL                                   */
    int64_t *order;             /* (P) argsort of the generation's fitness                             */
    void *state;                /* sx_cma_state                                                       */
    double *pen_ws;             /* constraints="Penalize" on the device, or NULL: as sx_cma_args.pen_ws (the covariance
                                 * diagonal is that of D (I + v v^T) D, vdcma/_vdcma.py:249-254)        */
    int64_t *pen_order;         /* (P), with pen_ws                                                   */
    int64_t P;
    int64_t hist_rows;
    int32_t n, mu, fun_id, maxiter, ilim, pad_;
    double cs, ds, cc, c1, cmu, mueff, wsum, xtol, ftol, insigma;
    uint32_t key0, key1;
} sx_vd_args;

int sx_vdcma_generation(const sx_vd_args *a, int64_t gen, void *stream);
/* two steps around one all-gather for candidates sharded over ranks, as sx_cmaes_generation_stage (stage 0 fills
 * ary_loc / arx_loc (rows,n) and fit_loc (rows) for rows [row0, row0 + rows); the injected pair is global rows 0, 1).
 * Stage 1 of a generation follows stage 0 of the SAME generation on the same stream: for wide models stage 0 also zeroes the
 * barrier words (behind t_k in Z) of the one-launch model update that stage 1 enqueues. */
int sx_vdcma_generation_stage(const sx_vd_args *a, int64_t gen, int stage, int64_t row0, int64_t rows, double *ary_loc,
                              double *arx_loc, double *fit_loc, void *stream);

/* ------------------------------------------------------------------------- *
 * Neighbourhood Algorithm: the resampling walk (csrc/sx_na.hip)
 * replaces na/_na.py:265-305 mutation(), everything that is O(popsize * models * ndim):
 *   sx_na_begin    X[i] = model kidx[i];  d2[i][m] = ((U[:, 1:] - X[i, 1:]) ** 2).sum(axis=1)  (:277-280, numpy's
 *                  pairwise order) for every stored model m
 *   sx_na_axis     axis step j of all walks: the pending `d2 += (U[:, jp] - X[i, jp]) ** 2 - (U[:, jp+1] - X[i, jp+1]) ** 2`
 *                  of the previous free axis jp (-1: none), lim (:288), low / high (:290-294),
 *                  X[i, j] = low + (high - low) * u[i, j] (:296: np.random.uniform(low, high)); xnew[i] = X[i, j]
 *   sx_na_commit   fixed axes of the finished samples -> 0 (:283-286) and the samples appended to the model store
 *   sx_na_uniforms the generation's uniforms from the device generator (Philox, rng="philox")
 * The scalar recurrence d1 (:298-300: numpy SCALAR `** 2` = libm pow) is evaluated by the caller between two
 * axis steps and handed in as d1[i].  Models are stored column-major: XT[l * cap + m]; kidx DEVICE int64 (P) store
 * positions; X DEVICE (P,n) the samples being built; d2 DEVICE (P,cap); u DEVICE (P,n) doubles in [0,1);
 * ws DEVICE scratch of 2 * P * sx_na_blocks(M) doubles; fixed DEVICE int32 (n), 1 where upper == lower.
 * ------------------------------------------------------------------------- */
int sx_na_blocks(int64_t M);
int sx_na_begin(const double *XT, int64_t cap, int64_t M, int n, const int64_t *kidx, int64_t P, double *X, double *d2,
                void *stream);
int sx_na_axis(const double *XT, int64_t cap, int64_t M, int n, int j, int jp, const int64_t *kidx, int64_t P,
               const double *u, const double *d1, double *X, double *d2, double *ws, double *xnew, void *stream);
int sx_na_commit(double *X, int64_t P, int n, const int32_t *fixed, double *XT, int64_t cap, int64_t M, void *stream);
int sx_na_uniforms(double *U, int64_t P, int n, uint32_t gen, uint32_t key0, uint32_t key1, void *stream);

/* ------------------------------------------------------------------------- *
 * numpy-legacy random stream (host): bit-exact MT19937 replica of what the
 * reference draws through np.random.* after np.random.seed(seed)
 * (de/_de.py:148-149, cpso/_cpso.py:153-154, cmaes/_cmaes.py:116-117;
 * algorithms: SURVEY.md Appendix A).  Host memory only.
 * ------------------------------------------------------------------------- */
typedef struct sx_mt sx_mt;
sx_mt *sx_mt_create(uint32_t seed);                       /* np.random.seed(seed)                 */
void sx_mt_destroy(sx_mt *g);
void sx_mt_seed(sx_mt *g, uint32_t seed);
void sx_mt_random(sx_mt *g, double *out, int64_t count);  /* rand / uniform(size=)               */
void sx_mt_uniform(sx_mt *g, double lo, double hi, double *out, int64_t count); /* uniform(lo,hi,size) */
void sx_mt_uniform_rows(sx_mt *g, const double *lo, const double *hi, int n, int64_t rows,
                        double *out);                     /* uniform(lo[n],hi[n],(rows,n))       */
void sx_mt_randn(sx_mt *g, double *out, int64_t count);   /* randn / normal(0,1) (polar, cached) */
void sx_mt_randint(sx_mt *g, int64_t high, int64_t *out, int64_t count); /* randint(high,size=)  */
void sx_mt_permutation(sx_mt *g, int64_t n, int64_t *out); /* permutation(n)                     */
/* _common.py:109-120 lhs: rand(P,n) then n x permutation(P); out[i][j] = (x[perm_j[i]][j]) * scale[j] + shift[j] with
 * x = rand / P + lin[i]; lin (P) = numpy's linspace(-1, 1, P, endpoint=False), scale / shift (n) = 0.5*(upper -/+ lower) */
void sx_mt_latin_hypercube(sx_mt *g, int64_t P, int n, const double *lin, const double *scale, const double *shift,
                           double *out);
/* de/_de.py:304-311 delete_shuffle_sync: P permutations of P-1, first k rows kept: donors[k][P] */
void sx_mt_de_donors(sx_mt *g, int64_t P, int k, int32_t *donors);
/* the per-individual draws of ONE de_async generation after r1 (de/_de.py:376-382): for i = 0..P-1
 * permutation(delete(arange(P), i)) -> donors[t*P+i] (t < k), randint(n) -> irand[i], and with
 * constraints="Random" uniform(lower, upper, n) -> resample[i*n ..] (resample NULL: not drawn) */
void sx_mt_de_async_draws(sx_mt *g, int64_t P, int k, int n, int32_t *donors, int32_t *irand, const double *lower,
                          const double *upper, double *resample);
/* interchange with np.random.get_state()/set_state(): key[624], pos, has_gauss, cached_gaussian */
void sx_mt_get_state(sx_mt *g, uint32_t *key, int *pos, int *has_gauss, double *gauss);
void sx_mt_set_state(sx_mt *g, const uint32_t *key, int pos, int has_gauss, double gauss);

#ifdef __cplusplus
}
#endif
#endif /* STOCHOPY_HIP_H */
