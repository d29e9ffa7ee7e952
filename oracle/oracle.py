"""ctypes front-end of the CPU oracle (oracle/kpilqr_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package `trajoptkp_amd`.

All arrays are float64 numpy in the reference's (Eigen) layout: one COLUMN-MAJOR matrix per
time-step, i.e. numpy shape [T, cols, rows] C-contiguous.  See kpilqr_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(native=False, out_dir=None):
    """Compile the oracle. native=True -> -march=native build for CPU-baseline timing."""
    out_dir = out_dir or os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    name = "libkpilqr_oracle_native.so" if native else "libkpilqr_oracle.so"
    out = os.path.join(out_dir, name)
    src = os.path.join(_HERE, "kpilqr_oracle.c")
    march = "native" if native else "x86-64-v3"
    cmd = ["gcc", "-O3", f"-march={march}", "-ffp-contract=off", "-fPIC", "-std=c99",
           "-shared", "-o", out, src, "-lm", "-lpthread"]
    subprocess.check_call(cmd)
    return out


def lib(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or os.path.join(_HERE, "_build", "libkpilqr_oracle.so")
    if not os.path.exists(p):
        p = build()
    L = C.CDLL(p)
    L.orc_fd_difference.argtypes = [C.c_int, C.c_int, C.c_int, _i, _i, _u8, _i, _d, _d, _d,
                                    C.c_double, _d, _d]
    L.orc_fd_difference.restype = None
    L.orc_kp_set_interval.argtypes = [C.c_int, C.c_int, C.c_int, _i, _i]
    L.orc_kp_set_interval.restype = C.c_int
    L.orc_kp_adaptive_jerk.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _d, C.c_double, _d, _i, _i]
    L.orc_kp_adaptive_jerk.restype = C.c_int
    L.orc_kp_adaptive_accel.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _d, _d, _i, _i]
    L.orc_kp_adaptive_accel.restype = C.c_int
    L.orc_kp_velocity_change.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _d, _d, _i, _i]
    L.orc_kp_velocity_change.restype = C.c_int
    L.orc_kp_iterative_error.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _d, _i, _i]
    L.orc_kp_iterative_error.restype = C.c_int
    L.orc_kp_percentages.argtypes = [C.c_int, C.c_int, _i, _i, _d]
    L.orc_kp_percentages.restype = None
    L.orc_interpolate.argtypes = [C.c_int, C.c_int, C.c_int, _i, _i, _d, _d]
    L.orc_interpolate.restype = None
    L.orc_cost_function.argtypes = [C.c_int, _d, _d]
    L.orc_cost_function.restype = C.c_double
    L.orc_cost_derivs.argtypes = [C.c_int] * 4 + [_d] * 9
    L.orc_cost_derivs.restype = None
    L.orc_backward.argtypes = [C.c_int] * 3 + [_d] * 6 + [C.c_double, C.c_int, _d, _d,
                                                         C.POINTER(C.c_double)]
    L.orc_backward.restype = C.c_int
    L.orc_llt_is_pd.argtypes = [C.c_int, _d]
    L.orc_llt_is_pd.restype = C.c_int
    L.orc_ldlt_inverse.argtypes = [C.c_int, _d, _d]
    L.orc_ldlt_inverse.restype = None
    L.orc_alphas.argtypes = [C.c_int, _d]
    L.orc_alphas.restype = None
    L.orc_forward_linear.argtypes = [C.c_int] * 4 + [_d] * 11 + [_d, C.c_void_p]
    L.orc_forward_linear.restype = None
    L.orc_update_lambda.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_update_lambda.restype = C.c_int
    L.orc_check_convergence.argtypes = [C.c_double] * 3
    L.orc_check_convergence.restype = C.c_int
    L.orc_linesearch_accept.argtypes = [C.c_int, _d, C.c_double, C.POINTER(C.c_double),
                                        C.POINTER(C.c_int), C.POINTER(C.c_double),
                                        C.c_double, C.c_double]
    L.orc_linesearch_accept.restype = C.c_int
    L.orc_filter_dynamics.argtypes = [C.c_int, C.c_int, C.c_int, _d, C.c_int, _d]
    L.orc_filter_dynamics.restype = None
    L.orc_dof_importance.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _d, _d]
    L.orc_dof_importance.restype = None
    L.orc_alphas_svr.argtypes = [C.c_int, _d]
    L.orc_alphas_svr.restype = None
    L.orc_iteration.argtypes = [C.POINTER(Problem), _d, _d, C.POINTER(C.c_double), _d]
    L.orc_iteration.restype = C.c_int
    L.orc_iteration_batch.argtypes = [C.POINTER(Problem), C.c_int, C.c_int]
    L.orc_iteration_batch.restype = C.c_double
    if path is None:
        _LIB = L
    return L


class Problem(C.Structure):
    """struct orc_problem (kpilqr_oracle.h)."""
    _fields_ = [("dof", C.c_int), ("m", C.c_int), ("nr", C.c_int), ("T", C.c_int), ("njobs", C.c_int),
                ("pd_stride", C.c_int), ("n_alpha", C.c_int), ("eps", C.c_double), ("lam", C.c_double),
                ("job_t", C.c_void_p), ("job_col", C.c_void_p), ("job_nom", C.c_void_p), ("job_mode", C.c_void_p),
                ("xplus", C.c_void_p), ("xminus", C.c_void_p), ("xnom", C.c_void_p),
                ("kp_offs", C.c_void_p), ("kp_cols", C.c_void_p),
                ("r", C.c_void_p), ("r_x", C.c_void_p), ("r_u", C.c_void_p), ("w_run", C.c_void_p),
                ("w_term", C.c_void_p), ("u_nom", C.c_void_p), ("ctrl_lim", C.c_void_p)]


def make_c_problem(p, b, lam=None, pd_stride=100, n_alpha=6):
    """orc_problem for trajectory b of a synth problem dict; returns (struct, keep-alive list)."""
    sel = p["job_b"] == b
    arrs = dict(job_t=_c(p["job_t"][sel], np.int32), job_col=_c(p["job_col"][sel], np.int32),
                job_nom=_c(p["job_nom"][sel], np.int32), job_mode=_c(p["job_mode"][sel], np.uint8),
                xplus=_c(p["xplus"][sel]), xminus=_c(p["xminus"][sel]), xnom=_c(p["xnom"]),
                kp_offs=_c(p["kp_rows"][b][0], np.int32), kp_cols=_c(p["kp_rows"][b][1], np.int32),
                r=_c(p["r"][b]), r_x=_c(p["r_x"][b]), r_u=_c(p["r_u"][b]), w_run=_c(p["w_run"]), w_term=_c(p["w_term"]),
                u_nom=_c(p["u_nom"][b]), ctrl_lim=_c(p["ctrl_lim"]))
    s = Problem(p["dof"], p["m"], p["nr"], p["T"], int(sel.sum()), pd_stride, n_alpha, p["eps"],
                p["lam"] if lam is None else lam, **{k: v.ctypes.data for k, v in arrs.items()})
    return s, arrs


def iteration(p, b, **kw):
    s, keep = make_c_problem(p, b, **kw)
    n, m, T = p["n"], p["m"], p["T"]
    K = np.zeros((T, n, m)); k = np.zeros((T, m)); dJ = C.c_double(0.0); cost = np.zeros(16)
    st = lib().orc_iteration(C.byref(s), K, k, C.byref(dJ), cost)
    return st, K, k, dJ.value, cost[:s.n_alpha].copy()


def iteration_batch_seconds(p, b, nthreads, reps, **kw):
    s, keep = make_c_problem(p, b, **kw)
    return lib().orc_iteration_batch(C.byref(s), nthreads, reps)


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


# ---- key-points -------------------------------------------------------------------------
def _kp_out(dof, T, extra=0):
    return np.zeros(T + 1, np.int32), np.zeros(T * dof + dof + extra, np.int32)


def kp_set_interval(dof, T, min_N):
    offs, cols = _kp_out(dof, T)
    cnt = lib().orc_kp_set_interval(dof, T, min_N, offs, cols)
    return offs, cols[:cnt].copy()


def kp_adaptive_jerk(dof, T, min_N, max_N, thr, dt, X):
    offs, cols = _kp_out(dof, T, extra=T * dof)
    cnt = lib().orc_kp_adaptive_jerk(dof, T, min_N, max_N, _c(thr), dt, _c(X), offs, cols)
    return offs, cols[:cnt].copy()


def kp_adaptive_accel(dof, T, min_N, max_N, thr, X):
    offs, cols = _kp_out(dof, T, extra=T * dof)
    cnt = lib().orc_kp_adaptive_accel(dof, T, min_N, max_N, _c(thr), _c(X), offs, cols)
    return offs, cols[:cnt].copy()


def kp_velocity_change(dof, T, min_N, max_N, thr, X):
    offs, cols = _kp_out(dof, T)
    cnt = lib().orc_kp_velocity_change(dof, T, min_N, max_N, _c(thr), _c(X), offs, cols)
    return offs, cols[:cnt].copy()


def kp_iterative_error(dof, T, min_N, threshold, A):
    offs, cols = _kp_out(dof, T)
    cnt = lib().orc_kp_iterative_error(dof, T, min_N, threshold, _c(A), offs, cols)
    return offs, cols[:cnt].copy()


def kp_percentages(dof, T, offs, cols):
    pct = np.zeros(dof)
    lib().orc_kp_percentages(dof, T, _c(offs, np.int32), _c(cols, np.int32), pct)
    return pct


def kp_rows(offs, cols):
    """CSR -> the reference's vector<vector<int>>."""
    return [list(map(int, cols[offs[t]:offs[t + 1]])) for t in range(len(offs) - 1)]


# ---- a2 / a4 / a6 / a7 / a8 ---------------------------------------------------------------
def fd_difference(n, m, job_t, job_col, job_mode, job_nom, xplus, xminus, xnom, eps, A, B):
    """In place on A [T,n,n] and B [T,m,n] (column-major per step)."""
    nj = len(job_t)
    xn = _c(xnom) if xnom is not None and len(xnom) else np.zeros((1, n))
    lib().orc_fd_difference(n, m, nj, _c(job_t, np.int32), _c(job_col, np.int32),
                            _c(job_mode, np.uint8), _c(job_nom, np.int32),
                            _c(xplus), _c(xminus), xn, eps, A, B)


def interpolate(dof, m, T, offs, cols, A, B):
    """In place on A, B."""
    lib().orc_interpolate(dof, m, T, _c(offs, np.int32), _c(cols, np.int32), A, B)


def cost_derivs(n, m, nr, T, r, r_x, r_u, w_run, w_term):
    l_x = np.zeros((T, n)); l_xx = np.zeros((T, n, n)); l_u = np.zeros((T, m)); l_uu = np.zeros((T, m, m))
    lib().orc_cost_derivs(n, m, nr, T, _c(r), _c(r_x), _c(r_u), _c(w_run), _c(w_term),
                          l_x, l_xx, l_u, l_uu)
    return l_x, l_xx, l_u, l_uu


def cost_function(r, w):
    return lib().orc_cost_function(len(w), _c(r), _c(w))


def backward(n, m, T, A, B, l_x, l_xx, l_u, l_uu, lam, pd_stride=100):
    K = np.zeros((T, n, m)); k = np.zeros((T, m)); dJ = C.c_double(0.0)
    st = lib().orc_backward(n, m, T, _c(A), _c(B), _c(l_x), _c(l_xx), _c(l_u), _c(l_uu),
                            lam, pd_stride, K, k, C.byref(dJ))
    return st, K, k, dJ.value


def llt_is_pd(M):
    return bool(lib().orc_llt_is_pd(M.shape[0], _c(M.T)))


def ldlt_inverse(M):
    out = np.zeros_like(M, dtype=np.float64)
    lib().orc_ldlt_inverse(M.shape[0], _c(M.T), out)
    return out.T.copy()


def alphas(n_alpha=6):
    a = np.zeros(n_alpha)
    lib().orc_alphas(n_alpha, a)
    return a


def forward_linear(n, m, T, alphas_, A, B, K, k, l_x, l_xx, l_u, l_uu, u_nom, ctrl_lim, want_U=False):
    na = len(alphas_)
    cost = np.zeros(na)
    U = np.zeros((na, T, m)) if want_U else None
    lib().orc_forward_linear(n, m, T, na, _c(alphas_), _c(A), _c(B), _c(K), _c(k), _c(l_x), _c(l_xx),
                             _c(l_u), _c(l_uu), _c(u_nom), _c(ctrl_lim), cost,
                             U.ctypes.data_as(C.c_void_p) if want_U else None)
    return (cost, U) if want_U else cost


def alphas_svr(n_alpha=6):
    a = np.zeros(n_alpha)
    lib().orc_alphas_svr(n_alpha, a)
    return a


def filter_dynamics(dof, T, method, coefs, A):
    """A: [T][n][n] in the reference layout of this module (see interpolate); returns the filtered copy."""
    Ac = np.ascontiguousarray(A, np.float64).copy()
    co = np.ascontiguousarray(coefs, np.float64)
    lib().orc_filter_dynamics(dof, T, {"low_pass": 0, "FIR": 1}[method], co, len(co), Ac)
    return Ac


def dof_importance(dof, m, T, sampling, K):
    sums = np.zeros(dof)
    lib().orc_dof_importance(dof, m, T, sampling, np.ascontiguousarray(K, np.float64), sums)
    return sums


def update_lambda(lam, valid, factor=10.0, min_lambda=1e-4, max_lambda=10.0):
    l = C.c_double(lam)
    ex = lib().orc_update_lambda(C.byref(l), int(valid), factor, min_lambda, max_lambda)
    return l.value, bool(ex)


def check_convergence(old, new, eps=0.02):
    return bool(lib().orc_check_convergence(old, new, eps))


def linesearch_accept(costs, old_cost, lam, factor=10.0, max_lambda=10.0):
    nc = C.c_double(0.0); acc = C.c_int(0); l = C.c_double(lam)
    best = lib().orc_linesearch_accept(len(costs), _c(costs), old_cost, C.byref(nc), C.byref(acc),
                                       C.byref(l), factor, max_lambda)
    return best, nc.value, bool(acc.value), l.value


def dof_importance_svd(dof, m, T, sampling, K):
    """iLQR_SVR::LeastImportantDofs, singular-vector branch (src/Optimiser/iLQR_SVR.cpp:902-925), restated on numpy's
    SVD (LAPACK) in place of Eigen::JacobiSVD: K [T][n][m] column-major m x n per step.  Parity unpinned."""
    sums = np.zeros(dof)
    for t in range(0, T, sampling):
        Kt = np.asarray(K[t]).T                      # m x n
        _, S, Vt = np.linalg.svd(Kt, full_matrices=True)
        V = Vt.T
        for mm in range(min(3, m)):
            sums += np.abs(V[:dof, mm] * S[mm]) + np.abs(V[dof:2 * dof, mm] * S[mm])
    return sums / T
