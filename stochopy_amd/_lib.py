"""ctypes binding of the C ABI declared in include/stochopy_hip.h.

The product path has NO CPU fallback: if the shared library was not built, or
it cannot be loaded, importing a device op raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstochopy_hip.so")

SX_STATUS_NONE = 100
# rows of more than NARROW_DIM elements are "wide" (csrc/sx_wide.hip: one workgroup per row): no chained / peer-exchange /
# ordered-sweep kernels for those, everything else is served; WIDE_DIM is the longest row
# NARROW_DIM is what the wavefront-per-row kernels can serve (the limit of the ordered sweeps, the chained / peer-exchange
# kernels and full CMA-ES); from which length on rows actually TAKE the wide kernels is the library's choice: wide_from()
NARROW_DIM, WIDE_DIM = 4096, 262144


def wide_from():
    """Rows of more than this many elements are served by the one-workgroup-per-row kernels (csrc/sx_device.hpp kWideFrom)."""
    return int(lib().sx_wide_from())
SX_RNG_HOST, SX_RNG_PHILOX = 0, 1

FUN_IDS = {
    "ackley": 0,
    "griewank": 1,
    "quartic": 2,
    "rastrigin": 3,
    "rosenbrock": 4,
    "sphere": 5,
    "styblinski_tang": 6,
}
DE_STRATEGIES = {"rand1bin": 0, "rand2bin": 1, "best1bin": 2, "best2bin": 3}
DE_DONORS = {"rand1bin": 3, "rand2bin": 5, "best1bin": 2, "best2bin": 4}

vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int32
f64 = C.c_double


class SxState(C.Structure):
    _fields_ = [("it", i64), ("gbidx", i64), ("gfit", f64), ("dx", f64), ("status", i32), ("done", i32),
                ("reserved", i64 * 3)]


class SxDeArgs(C.Structure):
    _fields_ = [
        ("buf0", vp), ("buf1", vp), ("fit", vp), ("candfit", vp), ("gbest", vp), ("lower", vp),
        ("upper", vp), ("state", vp), ("part_f", vp), ("part_i", vp), ("r1", vp), ("donors", vp),
        ("irand", vp), ("resample", vp),
        ("P", i64), ("ld", i64), ("row0", i64),
        ("n", i32), ("fun_id", i32), ("strategy", i32), ("constraints", i32), ("rng", i32), ("maxiter", i32),
        ("F", f64), ("CR", f64), ("xtol", f64), ("ftol", f64),
        ("key0", C.c_uint32), ("key1", C.c_uint32),
    ]


class SxPsoArgs(C.Structure):
    _fields_ = [
        ("X", vp), ("V", vp), ("pbest", vp), ("pbestfit", vp), ("candfit", vp), ("gbest", vp), ("lower", vp),
        ("upper", vp), ("state", vp), ("part_f", vp), ("part_i", vp), ("r1", vp), ("r2", vp), ("pending_restart", vp),
        ("P", i64), ("ld", i64), ("row0", i64),
        ("n", i32), ("fun_id", i32), ("constraints", i32), ("rng", i32), ("maxiter", i32), ("pad", i32),
        ("w", f64), ("c1", f64), ("c2", f64), ("xtol", f64), ("ftol", f64),
        ("key0", C.c_uint32), ("key1", C.c_uint32),
    ]


SX_MAX_PEERS = 8
SX_IPC_HANDLE_BYTES = 64


class SxXchgArgs(C.Structure):
    _fields_ = [("peer", vp * SX_MAX_PEERS), ("world", i32), ("rank", i32), ("timeout_ticks", i64), ("error", vp),
                ("relay", vp), ("pop0", vp * SX_MAX_PEERS), ("pop1", vp * SX_MAX_PEERS), ("global_rows", i64),
                ("shard_rows", i64)]


class SxCmaState(C.Structure):
    _fields_ = [("it", i64), ("nfev", i64), ("best_row", i64), ("fbest", f64), ("sigma", f64), ("sigma_next", f64),
                ("tmp_coef", f64), ("psnorm", f64), ("status", i32), ("done", i32), ("stop_it", i64),
                ("reserved", f64 * 6)]


class SxVdArgs(C.Structure):
    _fields_ = [
        ("Z", vp), ("ary", vp), ("arx", vp), ("fit", vp), ("xmean", vp), ("xold", vp), ("dx", vp), ("dvec", vp),
        ("vvec", vp), ("vn", vp), ("pc", vp), ("zinj", vp), ("dy", vp), ("w", vp), ("mws", vp), ("mout", vp),
        ("besthist", vp), ("xm", vp), ("xstd", vp), ("xbest", vp), ("hist_x", vp), ("hist_f", vp), ("order", vp),
        ("state", vp), ("pen_ws", vp), ("pen_order", vp), ("P", i64), ("hist_rows", i64),
        ("n", i32), ("mu", i32), ("fun_id", i32), ("maxiter", i32), ("ilim", i32), ("pad", i32),
        ("cs", f64), ("ds", f64), ("cc", f64), ("c1", f64), ("cmu", f64), ("mueff", f64), ("wsum", f64), ("xtol", f64),
        ("ftol", f64), ("insigma", f64), ("key0", C.c_uint32), ("key1", C.c_uint32),
    ]


class SxCmaArgs(C.Structure):
    _fields_ = [
        ("Z", vp), ("arx", vp), ("fit", vp), ("xmean", vp), ("xold", vp), ("ps", vp), ("pc", vp), ("C", vp), ("B", vp),
        ("D", vp), ("eigw", vp), ("w", vp), ("Y", vp), ("part", vp), ("step", vp), ("isc", vp), ("xnew", vp),
        ("ypart", vp), ("besthist", vp), ("xm", vp), ("xstd", vp),
        ("xbest", vp), ("hist_x", vp), ("hist_f", vp), ("order", vp), ("state", vp), ("eigh_ws", vp),
        ("pen_ws", vp), ("pen_order", vp),
        ("eigh_ws_bytes", i64), ("P", i64), ("hist_rows", i64),
        ("n", i32), ("mu", i32), ("fun_id", i32), ("maxiter", i32), ("ilim", i32), ("eig_sweeps", i32),
        ("cs", f64), ("cc", f64), ("c1", f64), ("cmu", f64), ("damps", f64), ("chind", f64), ("mueff", f64),
        ("xtol", f64), ("ftol", f64), ("insigma", f64),
        ("key0", C.c_uint32), ("key1", C.c_uint32),
    ]


# name -> (restype, argtypes); every symbol include/stochopy_hip.h declares
PROTOTYPES = {
    "sx_abi_version": (C.c_int, []),
    "sx_wide_from": (C.c_int, []),
    "sx_set_wide_from": (C.c_int, [C.c_int]),
    "sx_last_error": (C.c_char_p, []),
    "sx_device_count": (C.c_int, []),
    "sx_struct_size": (C.c_int, [C.c_int]),
    "sx_sum_plan": (C.c_int, [i64, vp, C.c_int]),
    "sx_fun_terms": (i64, [C.c_int, C.c_int]),
    "sx_num_partials": (i64, [i64, C.c_int]),
    "sx_rows_per_workgroup": (C.c_int, [C.c_int]),
    "sx_eval": (C.c_int, [C.c_int, vp, i64, C.c_int, i64, vp, vp, vp, vp, vp, vp]),
    "sx_philox_lhs": (C.c_int, [vp, i64, C.c_int, i64, i64, i64, vp, vp, C.c_uint32, C.c_uint32, vp]),
    "sx_argmin": (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp]),
    "sx_de_generation": (C.c_int, [C.POINTER(SxDeArgs), C.c_int, vp]),
    "sx_select_finalize": (C.c_int, [vp, vp, i64, vp, vp, i64, C.c_int, vp, vp, C.c_int, f64, f64, vp]),
    "sx_shard_best": (C.c_int, [vp, vp, i64, vp, vp, i64, C.c_int, vp, i64, vp, vp]),
    "sx_gather_finalize": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, f64, f64, vp]),
    "sx_de_shard_generation": (C.c_int, [C.POINTER(SxDeArgs), vp, vp]),
    "sx_de_graph_create": (C.c_int, [C.POINTER(SxDeArgs), C.c_int, C.POINTER(vp)]),
    "sx_de_chain_launch": (C.c_int, [C.POINTER(SxDeArgs), C.c_int, C.c_int, vp]),
    "sx_de_chain_graph_create": (C.c_int, [C.POINTER(SxDeArgs), C.c_int, C.c_int, C.POINTER(vp)]),
    "sx_xchg_bytes": (i64, [C.c_int, C.c_int]),
    "sx_xchg_relay_bytes": (i64, [C.c_int]),
    "sx_xchg_alloc": (C.c_int, [i64, C.POINTER(vp), vp]),
    "sx_xchg_free": (C.c_int, [vp]),
    "sx_pop_alloc": (C.c_int, [i64, C.POINTER(vp), vp]),
    "sx_xchg_open": (C.c_int, [vp, C.POINTER(vp)]),
    "sx_xchg_close": (C.c_int, [vp]),
    "sx_xchg_probe": (C.c_int, [C.POINTER(SxXchgArgs), C.c_int, C.c_int, vp]),
    "sx_xchg_read_record": (C.c_int, [C.POINTER(SxXchgArgs), C.c_int, C.c_int, C.c_int, vp, vp]),
    "sx_xchg_finalize": (C.c_int, [vp, vp, i64, vp, vp, i64, C.c_int, i64, vp, vp, C.c_int, f64, f64,
                                   C.POINTER(SxXchgArgs), vp]),
    "sx_de_p2p_launch": (C.c_int, [C.POINTER(SxDeArgs), C.POINTER(SxXchgArgs), C.c_int, C.c_int, vp]),
    "sx_de_p2p_graph_create": (C.c_int, [C.POINTER(SxDeArgs), C.POINTER(SxXchgArgs), C.c_int, C.c_int,
                                         C.POINTER(vp)]),
    "sx_graph_launch": (C.c_int, [vp, vp]),
    "sx_graph_destroy": (C.c_int, [vp]),
    "sx_pso_generation": (C.c_int, [C.POINTER(SxPsoArgs), C.c_int, vp]),
    "sx_pso_chain_supported": (C.c_int, [C.POINTER(SxPsoArgs)]),
    "sx_pso_chain_launch": (C.c_int, [C.POINTER(SxPsoArgs), vp, C.c_int, C.c_int, vp]),
    "sx_pso_chain_graph_create": (C.c_int, [C.POINTER(SxPsoArgs), vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "sx_pso_radius": (C.c_int, [C.POINTER(SxPsoArgs), vp, vp]),
    "sx_pso_restart_select": (C.c_int, [C.POINTER(SxPsoArgs), vp, f64, f64, vp, vp]),
    "sx_pso_restart_apply": (C.c_int, [C.POINTER(SxPsoArgs), vp, vp, vp, i64, vp]),
    "sx_pso_restart_select_gathered": (C.c_int, [C.POINTER(SxPsoArgs), vp, C.c_int, f64, f64, vp, vp]),
    "sx_pso_graph_create": (C.c_int, [C.POINTER(SxPsoArgs), C.c_int, vp, f64, f64, vp, C.POINTER(vp)]),
    "sx_de_async_generation": (C.c_int, [C.POINTER(SxDeArgs), vp]),
    "sx_de_propose": (C.c_int, [C.POINTER(SxDeArgs), vp, vp]),
    "sx_pso_move": (C.c_int, [C.POINTER(SxPsoArgs), vp]),
    "sx_rows_select": (C.c_int, [vp, i64, vp, vp, vp, i64, vp, vp, i64, C.c_int, vp, vp, vp, vp]),
    "sx_pso_async_generation": (C.c_int, [C.POINTER(SxPsoArgs), vp]),
    "sx_cmaes_sample": (C.c_int, [vp, f64, vp, vp, vp, vp, i64, C.c_int, vp]),
    "sx_cmaes_recombine": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "sx_cmaes_rank_mu": (C.c_int, [vp, vp, vp, C.c_int, vp, f64, vp, f64, f64, f64, vp, vp, C.c_int, vp]),
    "sx_cmaes_normals": (C.c_int, [vp, i64, C.c_int, i64, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "sx_symmetrize_upper": (C.c_int, [vp, C.c_int, vp]),
    "sx_cmaes_eval_penalized": (C.c_int, [C.c_int, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp]),
    "sx_vdcma_sample": (C.c_int, [vp, i64, C.c_int, i64, vp, vp, f64, vp, f64, vp, vp, vp, vp]),
    "sx_na_blocks": (C.c_int, [i64]),
    "sx_na_begin": (C.c_int, [vp, i64, i64, C.c_int, vp, i64, vp, vp, vp]),
    "sx_na_axis": (C.c_int, [vp, i64, i64, C.c_int, C.c_int, C.c_int, vp, i64, vp, vp, vp, vp, vp, vp, vp]),
    "sx_na_commit": (C.c_int, [vp, i64, C.c_int, vp, vp, i64, i64, vp]),
    "sx_na_uniforms": (C.c_int, [vp, i64, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "sx_vdcma_moments": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, f64, vp, vp, vp]),
    "sx_cmaes_generation": (C.c_int, [C.POINTER(SxCmaArgs), i64, C.c_int, vp]),
    "sx_cmaes_generation_stage": (C.c_int, [C.POINTER(SxCmaArgs), i64, C.c_int, C.c_int, i64, i64, vp, vp, vp]),
    "sx_cmaes_generation_phased": (C.c_int, [C.POINTER(SxCmaArgs), i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "sx_eigh_rounds_per_sweep": (C.c_int, [C.c_int]),
    "sx_vdcma_generation": (C.c_int, [C.POINTER(SxVdArgs), i64, vp]),
    "sx_vdcma_generation_stage": (C.c_int, [C.POINTER(SxVdArgs), i64, C.c_int, i64, i64, vp, vp, vp, vp]),
    "sx_eigh_workspace_bytes": (i64, [C.c_int]),
    "sx_eigh": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, i64, C.c_int, f64, vp]),
    "sx_eigh_refined": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, i64, C.c_int, f64, C.c_int, vp]),
    "sx_eigh_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(f64), vp]),
    "sx_eigh_set_refine": (C.c_int, [C.c_int]),
    "sx_eigh_set_flow": (C.c_int, [C.c_int]),
    "sx_mt_create": (vp, [C.c_uint32]),
    "sx_mt_destroy": (None, [vp]),
    "sx_mt_seed": (None, [vp, C.c_uint32]),
    "sx_mt_random": (None, [vp, vp, i64]),
    "sx_mt_uniform": (None, [vp, f64, f64, vp, i64]),
    "sx_mt_uniform_rows": (None, [vp, vp, vp, C.c_int, i64, vp]),
    "sx_mt_randn": (None, [vp, vp, i64]),
    "sx_mt_randint": (None, [vp, i64, vp, i64]),
    "sx_mt_permutation": (None, [vp, i64, vp]),
    "sx_mt_latin_hypercube": (None, [vp, i64, C.c_int, vp, vp, vp, vp]),
    "sx_mt_de_donors": (None, [vp, i64, C.c_int, vp]),
    "sx_mt_de_async_draws": (None, [vp, i64, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "sx_mt_get_state": (None, [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(f64)]),
    "sx_mt_set_state": (None, [vp, vp, C.c_int, C.c_int, f64]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises HipLibraryError if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C stochopy_amd/csrc`.  stochopy_amd has no CPU fallback.")
    try:
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if handle.sx_abi_version() != 1:
        raise HipLibraryError("ABI version mismatch; rebuild the library")
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise HipLibraryError(f"{what} failed (rc={rc}): {lib().sx_last_error().decode()}")
