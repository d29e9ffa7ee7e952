"""ctypes access to libkpilqr_host.so: the C++ host classes (KeypointGenerator, Differentiator,
iLQR_GPU -- trajoptkp_amd/host/) through the small C entry points of host_capi.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libkpilqr_host.so")
_host = None


def build_host():
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host")])
    return HOST_LIB_PATH


def load_host():
    global _host
    if _host is not None:
        return _host
    _lib.load()                       # libkpilqr.so (and torch's HIP runtime) first
    if not os.path.exists(HOST_LIB_PATH):
        build_host()
    H = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    H.kpilqr_host_keypoints.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_double, C.c_double,
                                        vp, vp, vp, vp, vp]
    H.kpilqr_host_keypoints.restype = C.c_int
    H.kpilqr_host_run_acrobot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_double, vp, C.c_int, vp, vp, vp]
    H.kpilqr_host_run_acrobot.restype = C.c_int
    H.kpilqr_host_fd_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    H.kpilqr_host_fd_bench.restype = C.c_int
    H.kpilqr_host_save_trajec.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    H.kpilqr_host_save_keypoints.argtypes = [C.c_char_p, C.c_int, vp, vp]
    H.kpilqr_host_task_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp]
    H.kpilqr_host_save_summary.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp]
    H.kpilqr_host_run_acrobot_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    _host = H
    return H


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def keypoints(method, dof, T, min_N, max_N=1, thresholds=None, iterative_error_threshold=0.0, dt=0.01, X=None, A=None):
    """KeypointGenerator::GenerateKeyPoints -> (offs[T+1], cols, percentages), CSR over time."""
    H = load_host()
    offs = np.zeros(T + 1, np.int32)
    cols = np.zeros(2 * T * dof + dof, np.int32)
    pct = np.zeros(dof)
    th = None if thresholds is None else np.ascontiguousarray(thresholds, np.float64)
    Xc = None if X is None else np.ascontiguousarray(X, np.float64)
    Ac = None if A is None else np.ascontiguousarray(A, np.float64)
    cnt = H.kpilqr_host_keypoints(method.encode(), dof, T, min_N, max_N, _p(th), iterative_error_threshold, dt,
                                  _p(Xc), _p(Ac), _p(offs), _p(cols), _p(pct))
    return offs, cols[:cnt].copy(), pct


def run_acrobot(T=100, min_N=5, max_iter=5, min_iter=2, method="set_interval", torque_weight=-1.0):
    H = load_host()
    hist = np.zeros(max_iter + 2); U = np.zeros(T); K0 = np.zeros(4); tm = np.zeros(4)
    it = H.kpilqr_host_run_acrobot(T, min_N, max_iter, min_iter, method.encode(), float(torque_weight), _p(hist), len(hist), _p(U), _p(K0), _p(tm))
    if it < 0:
        raise RuntimeError(f"kpilqr_host_run_acrobot failed: {it}")
    return dict(iterations=it, cost_history=hist[:it + 1].copy(), U=U, K0=K0,
                timings_ms=dict(derivs=tm[0], backward=tm[1], forward=tm[2], total=tm[3]))


def fd_bench(T=3000, reps=5, mode=1, fd_threads=16):
    """Host FD harness microbenchmark (no GPU): mode 0 = threads spawned per call + merge (the reference's
    shape), mode 1 = persistent pool writing in place.  Returns dict(seconds, columns, columns_per_s, pool, checksums)."""
    H = load_host()
    sec = C.c_double(0.0); cols = C.c_long(0); chk = np.zeros(2)
    pool = H.kpilqr_host_fd_bench(T, reps, mode, fd_threads, C.byref(sec), C.byref(cols), _p(chk))
    return dict(seconds=sec.value, columns=cols.value, columns_per_s=cols.value / max(sec.value, 1e-12), pool=pool,
                checksum_set=chk[0], checksum_order=chk[1])


def save_trajec(root_dir, A, B, X, U):
    """FileHandler::SaveTrajecInformation.  A [T][n][n], B [T][m][n] in the ABI's column-major-per-matrix layout
    (i.e. A[t, col, row]), X [T][n], U [T][m]."""
    H = load_host()
    T, n = X.shape; m = U.shape[1]
    a, b, x, u = (np.ascontiguousarray(v, np.float64) for v in (A, B, X, U))
    return H.kpilqr_host_save_trajec(root_dir.encode(), T, n // 2, m, _p(a), _p(b), _p(x), _p(u))


def save_keypoints(root_dir, offs, cols):
    H = load_host()
    o = np.ascontiguousarray(offs, np.int32); c = np.ascontiguousarray(cols, np.int32)
    return H.kpilqr_host_save_keypoints(root_dir.encode(), len(o) - 1, _p(o), _p(c))


def save_task(filename, start, targets):
    H = load_host()
    s = np.ascontiguousarray(start, np.float64); g = np.ascontiguousarray(targets, np.float64)
    return H.kpilqr_host_task_file(filename.encode(), 1, len(s), len(g), _p(s), _p(g))


def load_task(filename, n_start, n_targets):
    H = load_host()
    s = np.zeros(n_start); g = np.zeros(n_targets)
    rc = H.kpilqr_host_task_file(filename.encode(), 0, n_start, n_targets, _p(s), _p(g))
    return (s, g) if rc == 0 else None


def save_summary(filename, rows, timings):
    H = load_host()
    r = np.ascontiguousarray(rows, np.float64); t = np.ascontiguousarray(timings, np.float64)
    return H.kpilqr_host_save_summary(filename.encode(), r.shape[0], t.shape[2], _p(r), _p(t))


def run_acrobot_batch(q0s, T=100, min_N=5, max_iter=6, min_iter=2, torque_weight=-1.0, fused=False):
    """B acrobot swing-ups from the starts q0s [B][2] through ONE batched context (iLQR_GPU_Batch)."""
    H = load_host()
    q = np.ascontiguousarray(q0s, np.float64); B = q.shape[0]
    cap = max_iter + 2
    hist = np.zeros((B, cap)); its = np.zeros(B, np.int32); U = np.zeros((B, T)); stats = np.zeros(8)
    rc = H.kpilqr_host_run_acrobot_batch(B, T, min_N, max_iter, min_iter, float(torque_weight), _p(q), int(fused), _p(hist), cap, _p(its), _p(U), _p(stats))
    if rc < 0:
        raise RuntimeError(f"kpilqr_host_run_acrobot_batch failed: {rc}")
    return dict(iterations=its, cost_history=[hist[b][hist[b] >= 0] for b in range(B)], U=U, stats=stats)
