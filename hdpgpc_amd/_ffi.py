"""ctypes binding of libhdpgpc_hip.so (include/hdpgpc_hip.h).

The product path has no CPU fallback: importing this module without the built library raises.
"""
import ctypes
import os

import torch  # noqa: F401  (first: the library must bind to the HIP runtime PyTorch-ROCm already loaded, not a second copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
# HGP_LIB selects another build of the same library (only the diagnostic `make stamps` build uses it)
LIB_PATH = os.environ.get("HGP_LIB") or os.path.join(_HERE, "lib", "libhdpgpc_hip.so")

c_dp = ctypes.c_void_p  # device pointers travel as integers


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). hdpgpc_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    i32, i64, f64, vp, sz = ctypes.c_int, ctypes.c_long, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
    sigs = {
        "hgp_abi_version": (i32, []),
        "hgp_debug_mfma_f64": (i32, [vp, vp, vp, vp]),
        "hgp_debug_exp_neg_f64": (i32, [vp, i32, vp, vp]),
        "hgp_gram_rbf_f64": (i32, [vp, i32, vp, i32, f64, f64, f64, vp, vp]),
        "hgp_potrf_batched_f64": (i32, [vp, i32, i32, f64, f64, vp, vp, vp, vp]),
        "hgp_chol_inverse_batched_f64": (i32, [vp, i32, i32, f64, f64, vp, vp, vp]),
        "hgp_chol_inverse_ws_f64": (i32, [vp, i32, i32, f64, f64, vp, vp, vp, vp]),
        "hgp_rts_chain_f64": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
        "hgp_lds_chain_scatter_f64": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
        "hgp_add_diag_mean_f64": (i32, [vp, vp, i32, i32, f64, vp, vp]),
        "hgp_gemm_add_batched_f64": (i32, [i32, i32, i32, i32, i32, f64, vp, i32, i64, vp, i32, i64, f64, vp, i32, i64, vp, i32, i64, i32, vp]),
        "hgp_lds_chain_gather_f64": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i64, vp, vp]),
        "hgp_lds_chain_finish_f64": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "hgp_score_groups_f64": (i32, [vp, i32, vp, i64, vp, i64, i32, vp, vp, vp, vp, vp, i32, vp, f64, vp, vp, vp, vp]),
        "hgp_score_each_f64": (i32, [vp, i32, vp, i64, vp, i64, i32, vp, vp, vp, i32, f64, i32, vp, vp, vp, vp]),
        "hgp_pairs_plan_device_bytes": (sz, [i32, i32, i32]),
        "hgp_pairs_plan_create": (i32, [ctypes.POINTER(vp), i32, i32, i32, ctypes.POINTER(f64), vp, sz]),
        "hgp_pairs_plan_destroy": (None, [vp]),
        "hgp_pairs_plan_update": (i32, [vp, vp, vp, vp, vp, vp]),
        "hgp_pairs_plan_scalars": (vp, [vp]),
        "hgp_pairs_plan_set_accuracy": (i32, [vp, f64]),
        "hgp_pairs_plan_set_score_output": (i32, [vp, i32]),
        "hgp_loglik_pairs_f64": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
        "hgp_gemm_batched_f64": (i32, [i32, i32, i32, i32, i32, f64, vp, i32, i64, vp, i32, i64, f64, vp, i32, i64, i32, vp]),
        "hgp_matrix_lik_ws_bytes": (sz, [i32, i32]),
        "hgp_lat_error_f64": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
        "hgp_mniw_loglik_f64": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, sz, vp]),
        "hgp_warp_cov_f64": (i32, [vp, i32, f64, f64, f64, i32, vp, vp]),
        "hgp_chol_rank1_f64": (i32, [vp, vp, vp, vp, i32, i32, vp, vp]),
        "hgp_trsv_lower_quad_f64": (i32, [vp, i32, vp, i32, vp, vp]),
        "hgp_hmm_messages_f64": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
        "hgp_hmm_local_terms_f64": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "hgp_loglik_rows_f64": (i32, [vp, i32, i32, vp, vp, vp]),
        "hgp_assign_f64": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "hgp_warp_batch_f64": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, i32, f64, f64, f64, f64, vp, vp, vp, vp, vp, vp, vp]),
        "hgp_gemm_list_f64": (i32, [vp, i32, i32, vp]),
        "hgp_gemm_list_mapped_f64": (i32, [vp, i32, vp, i32, vp]),
        "hgp_chol_inverse_rhs_batched_f64": (i32, [vp, i32, i32, f64, f64, vp, vp, vp, i32, vp, vp, vp]),
        "hgp_lds_chain_gather2_f64": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i64, vp, vp, vp, vp]),
        "hgp_lds_chain_finish2_f64": (i32, [i32] + [vp] * 22 + [i32, vp, vp]),
        "hgp_copy_list_f64": (i32, [vp, i32, i64, vp]),
        "hgp_lds_chain_gather2_batched_f64": (i32, [vp, i32, i32, vp]),
        "hgp_lds_chain_finish2_batched_f64": (i32, [vp, i32, i32, vp]),
        "hgp_trsv_lower_solve_f64": (i32, [vp, i32, vp, i32, vp, vp, vp]),
        "hgp_lml_grad_f64": (i32, [vp, vp, vp, i32, f64, f64, f64, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sigs)


lib, EXPORTS = _load()


class GemmItem(ctypes.Structure):
    """hgp_gemm_item of include/hdpgpc_hip.h."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("D", ctypes.c_void_p), ("C", ctypes.c_void_p), ("C2", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int),
                ("ldc", ctypes.c_int), ("ldd", ctypes.c_int), ("tA", ctypes.c_int), ("tB", ctypes.c_int),
                ("alpha", ctypes.c_double), ("beta", ctypes.c_double), ("add_eye", ctypes.c_double)]


class ChainGatherDesc(ctypes.Structure):
    """hgp_chain_gather_desc of include/hdpgpc_hip.h."""
    _fields_ = [("st", ctypes.c_void_p * 8), ("pos", ctypes.c_void_p), ("out", ctypes.c_void_p), ("Y", ctypes.c_void_p),
                ("y_out", ctypes.c_void_p), ("W", ctypes.c_void_p), ("Rp", ctypes.c_void_p), ("y_row0", ctypes.c_long),
                ("T", ctypes.c_int)]


class ChainFinishDesc(ctypes.Structure):
    """hgp_chain_finish_desc of include/hdpgpc_hip.h."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("f_post", "c_post", "f_sm_prev", "P_sm_prev", "y", "part", "Snew", "info1", "info2", "W",
                                               "n0", "Nf", "bad_count", "stA", "stG", "stC", "stS", "stF", "stFsm", "stP", "stPsm", "pos",
                                               "sync")] + [("T", ctypes.c_int), ("annealing", ctypes.c_int)]


class HgpError(RuntimeError):
    pass


def check(rc, what):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError(f"{what}: bad argument")
    if rc == -2:
        raise NotImplementedError(f"{what}: size not supported by this build")
    raise HgpError(f"{what}: HIP error {rc - 1000}")
