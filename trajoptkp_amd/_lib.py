"""ctypes binding of libkpilqr.so -- the C ABI declared in include/kpilqr.h.

This is the only way Python reaches the engine: there is no Python/torch compute fallback.  If the
shared library is missing it is built in-tree with hipcc (trajoptkp_amd/csrc/Makefile); if that
fails, importing the engine raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPILQR_LIB: another build of the same library (kernel A/B runs, tools/build_variant.sh); the product default is in-tree
LIB_PATH = os.environ.get("KPILQR_LIB") or os.path.join(_HERE, "lib", "libkpilqr.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/kpilqr.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "kpilqr_create", "kpilqr_destroy", "kpilqr_version", "kpilqr_strerror", "kpilqr_get_dims",
    "kpilqr_host_alloc", "kpilqr_host_free", "kpilqr_sync", "kpilqr_device_ptr",
    "kpilqr_set_keypoints", "kpilqr_upload_fd", "kpilqr_fd_difference", "kpilqr_interpolate",
    "kpilqr_upload_residuals", "kpilqr_cost_derivs", "kpilqr_trajectory_cost",
    "kpilqr_backward", "kpilqr_download_gains", "kpilqr_upload_nominal", "kpilqr_forward_linear",
    "kpilqr_iterate", "kpilqr_set_AB", "kpilqr_get_AB", "kpilqr_set_cost_derivs",
    "kpilqr_get_cost_derivs", "kpilqr_backward_variant", "kpilqr_forward_variant",
    "kpilqr_upload_states", "kpilqr_generate_keypoints", "kpilqr_get_keypoints",
    "kpilqr_filter_dynamics", "kpilqr_dof_importance",
    "kpilqr_comm_unique_id", "kpilqr_comm_init", "kpilqr_allreduce_linesearch",
    "kpilqr_fd_slab_layout", "kpilqr_upload_fd_slab", "kpilqr_iterate_streamed", "kpilqr_resize",
    "kpilqr_keypoint_error_test", "kpilqr_fd_kp_layout", "kpilqr_upload_fd_kp", "kpilqr_backward_stats",
    "kpilqr_upload_kp_columns", "kpilqr_upload_residual_jacobians_const", "kpilqr_last_launch",
]


class Dims(C.Structure):
    _fields_ = [("dof", C.c_int), ("m", C.c_int), ("T", C.c_int), ("nr", C.c_int),
                ("batch", C.c_int), ("n_alpha", C.c_int), ("device", C.c_int), ("flags", C.c_int)]


class FdLayout(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("xplus", "xminus", "xnom", "job_b", "job_t", "job_col", "job_nom",
                                          "job_mode", "bytes")]


class FdkpLayout(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("entry_stride", "xplus", "xminus", "mode", "bytes", "elem_stride")]


class StreamIO(C.Structure):
    _fields_ = [("fd_slab", C.c_void_p), ("njobs", C.c_int), ("nnom", C.c_int),
                ("traj_job_first", C.c_void_p), ("traj_nom_first", C.c_void_p), ("eps", C.c_double),
                ("r", C.c_void_p), ("r_x", C.c_void_p), ("r_u", C.c_void_p), ("u_nom", C.c_void_p), ("lam", C.c_void_p),
                ("K", C.c_void_p), ("k", C.c_void_p), ("cost_pred", C.c_void_p), ("delta_J", C.c_void_p),
                ("status", C.c_void_p), ("fd_kp_slab", C.c_void_p), ("entries", C.c_int), ("kp_columns", C.c_void_p)]


ABI_MAJOR = 4                  # KPILQR_VERSION / 100 of the include/kpilqr.h this binding mirrors
FLAG_GENERIC_KERNELS = 1
FLAG_TILED_KERNELS = 2
FLAG_FUSED = 4

BUF_STEP_RECORDS, BUF_K, BUF_k, BUF_RESIDUALS, BUF_R_X, BUF_R_U, BUF_U_NOM, BUF_FD_XPLUS, \
    BUF_FD_XMINUS, BUF_COST_PRED, BUF_DELTA_J, BUF_STATUS = range(12)

ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_STATE = -1, -2, -3, -4, -5

_lib = None


def build(force=False):
    """Compile libkpilqr.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC, "-j4"] + (["-B"] if force else []))
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64
    # (torch/lib).  If libkpilqr.so were loaded first it would pull /opt/rocm's copy and a later
    # `import torch` would bring a SECOND runtime that finds no GPU.  Importing torch first makes the
    # loader resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.  torch is only
    # plumbing here (streams, events, torch.distributed); C/C++ hosts link the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.kpilqr_create.argtypes = [C.POINTER(Dims), vp, C.POINTER(vp)]
    L.kpilqr_destroy.argtypes = [vp]; L.kpilqr_destroy.restype = None
    L.kpilqr_version.argtypes = []
    L.kpilqr_strerror.argtypes = [vp]; L.kpilqr_strerror.restype = C.c_char_p
    L.kpilqr_get_dims.argtypes = [vp, C.POINTER(Dims)]
    L.kpilqr_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.kpilqr_host_free.argtypes = [vp, vp]
    L.kpilqr_sync.argtypes = [vp]
    L.kpilqr_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.kpilqr_set_keypoints.argtypes = [vp, vp, vp]
    L.kpilqr_upload_fd.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_double]
    L.kpilqr_fd_difference.argtypes = [vp]
    L.kpilqr_interpolate.argtypes = [vp]
    L.kpilqr_upload_residuals.argtypes = [vp, vp, vp, vp, vp, vp]
    L.kpilqr_cost_derivs.argtypes = [vp]
    L.kpilqr_trajectory_cost.argtypes = [vp, vp]
    L.kpilqr_backward.argtypes = [vp, vp, C.c_int, vp, vp]
    L.kpilqr_download_gains.argtypes = [vp, vp, vp]
    L.kpilqr_upload_nominal.argtypes = [vp, vp, vp]
    L.kpilqr_forward_linear.argtypes = [vp, vp, vp, vp]
    L.kpilqr_iterate.argtypes = [vp, vp, C.c_int, vp]
    L.kpilqr_set_AB.argtypes = [vp, vp, vp]
    L.kpilqr_get_AB.argtypes = [vp, vp, vp]
    L.kpilqr_set_cost_derivs.argtypes = [vp, vp, vp, vp, vp]
    L.kpilqr_get_cost_derivs.argtypes = [vp, vp, vp, vp, vp]
    L.kpilqr_backward_variant.argtypes = [vp]; L.kpilqr_backward_variant.restype = C.c_char_p
    L.kpilqr_forward_variant.argtypes = [vp]; L.kpilqr_forward_variant.restype = C.c_char_p
    L.kpilqr_upload_states.argtypes = [vp, vp]
    L.kpilqr_generate_keypoints.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp, C.c_double]
    L.kpilqr_get_keypoints.argtypes = [vp, vp, vp, C.c_int]
    L.kpilqr_filter_dynamics.argtypes = [vp, C.c_char_p, vp, C.c_int]
    L.kpilqr_dof_importance.argtypes = [vp, C.c_int, vp]
    L.kpilqr_comm_unique_id.argtypes = [vp]
    L.kpilqr_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.kpilqr_allreduce_linesearch.argtypes = [vp, vp]
    L.kpilqr_fd_slab_layout.argtypes = [vp, C.c_int, C.c_int, C.POINTER(FdLayout)]
    L.kpilqr_upload_fd_slab.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double]
    L.kpilqr_iterate_streamed.argtypes = [vp, C.POINTER(StreamIO), C.c_int, C.c_int]
    L.kpilqr_resize.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.kpilqr_keypoint_error_test.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, vp]
    L.kpilqr_fd_kp_layout.argtypes = [vp, C.c_int, C.POINTER(FdkpLayout)]
    L.kpilqr_upload_fd_kp.argtypes = [vp, vp, C.c_int, C.c_double]
    L.kpilqr_backward_stats.argtypes = [vp, C.c_int, vp]
    L.kpilqr_upload_kp_columns.argtypes = [vp, vp, C.c_int]
    L.kpilqr_upload_residual_jacobians_const.argtypes = [vp, vp, vp]
    L.kpilqr_last_launch.argtypes = [vp, C.c_int]; L.kpilqr_last_launch.restype = C.c_char_p
    for s in SYMBOLS:
        getattr(L, s)          # raises AttributeError if the .so lacks a declared symbol
    # the structs above (FdkpLayout, StreamIO ...) are those of ABI major version ABI_MAJOR: a library of another major version
    # would write past them or read them wrongly (round-4 advisor: kpilqr_fdkp_layout grew a field between 3.x and 4.0)
    got = L.kpilqr_version()
    if got // 100 != ABI_MAJOR:
        raise RuntimeError(f"{LIB_PATH}: kpilqr_version() = {got}, this binding is written for {ABI_MAJOR}xx (include/kpilqr.h)")
    _lib = L
    return L
