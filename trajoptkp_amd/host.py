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
    H.kpilqr_host_run_acrobot_batch2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, C.c_int, C.c_char_p, vp, C.c_int, vp, vp, vp]
    H.kpilqr_host_dof_importance.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp]
    H.kpilqr_host_relocate_records.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, vp]
    H.kpilqr_host_model_info.argtypes = [C.c_char_p, vp, vp, vp]
    H.kpilqr_host_fd_kp_check.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp]
    H.kpilqr_host_model_op.argtypes = [C.c_char_p, C.c_int, vp, vp, vp, vp, C.c_double, C.c_int, vp, vp]
    H.kpilqr_host_model_fd.argtypes = [C.c_char_p, vp, vp, vp, C.c_int, vp, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp, vp]
    H.kpilqr_host_optimise.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, vp, vp, C.c_int, vp, vp]
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
    hist = np.zeros(max_iter + 2); U = np.zeros(T); K0 = np.zeros(4); tm = np.zeros(8)
    it = H.kpilqr_host_run_acrobot(T, min_N, max_iter, min_iter, method.encode(), float(torque_weight), _p(hist), len(hist), _p(U), _p(K0), _p(tm))
    if it < 0:
        raise RuntimeError(f"kpilqr_host_run_acrobot failed: {it}")
    return dict(iterations=it, cost_history=hist[:it + 1].copy(), U=U, K0=K0,
                timings_ms=dict(derivs=tm[0], backward=tm[1], forward=tm[2], total=tm[3]),
                constant_jacobian_uploads=int(tm[4]), per_step_jacobian_uploads=int(tm[5]), last_backward_rxc=bool(tm[6]))


def fd_kp_check(model, T, min_N, stagger, u):
    """Differentiator::DynamicsDerivativesKp (key-point ordered payload) against the job-list fill on a rolled-out trajectory
    of a stand-in model; -> dict(mismatches, jobs, one_sided, entries)."""
    H = load_host()
    u = np.ascontiguousarray(u, np.float64)
    stats = np.zeros(3, np.int32)
    bad = H.kpilqr_host_fd_kp_check(model.encode(), int(T), int(min_N), int(stagger), _p(u), _p(stats))
    if bad < 0:
        raise RuntimeError(f"kpilqr_host_fd_kp_check failed: {bad}")
    return dict(mismatches=bad, jobs=int(stats[0]), one_sided=int(stats[1]), entries=int(stats[2]))


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


def relocate_records(slab, stride, B, dof, old_offs, new_offs, regen, in_place=True):
    """iLQR_GPU_Batch's record relocation on a byte slab (numpy uint8); returns the destination slab."""
    H = load_host()
    oo = np.ascontiguousarray(old_offs, np.int32); no = np.ascontiguousarray(new_offs, np.int32)
    rg = np.ascontiguousarray(regen, np.uint8)
    src = np.ascontiguousarray(slab, np.uint8)
    dst = src if in_place else np.zeros(max(int(no[-1]), 1) * stride, np.uint8)
    H.kpilqr_host_relocate_records(src.ctypes.data, None if in_place else dst.ctypes.data, stride, B, dof, _p(oo), _p(no), _p(rg))
    return dst


def run_acrobot_batch(q0s, T=100, min_N=5, max_iter=6, min_iter=2, torque_weight=-1.0, fused=False, method=None):
    """B acrobot swing-ups from the starts q0s [B][2] through ONE batched context (iLQR_GPU_Batch).  method: key-point
    method by name (None: set_interval)."""
    H = load_host()
    q = np.ascontiguousarray(q0s, np.float64); B = q.shape[0]
    cap = max_iter + 2
    hist = np.zeros((B, cap)); its = np.zeros(B, np.int32); U = np.zeros((B, T)); stats = np.zeros(8)
    rc = H.kpilqr_host_run_acrobot_batch2(B, T, min_N, max_iter, min_iter, float(torque_weight), _p(q), int(fused),
                                          None if method is None else method.encode(), _p(hist), cap, _p(its), _p(U), _p(stats))
    if rc < 0:
        raise RuntimeError(f"kpilqr_host_run_acrobot_batch failed: {rc}")
    return dict(iterations=its, cost_history=[hist[b][hist[b] >= 0] for b in range(B)], U=U, stats=stats)


# ---- stand-in models by name ("acrobot", "floating_body"): primitives for the oracle's restatement of the host FD loops ----
class Model:
    """The primitives of a stand-in simulator/task pair, one C call each (no GPU involved)."""

    def __init__(self, name):
        self.name = name.encode()
        H = load_host()
        info = np.zeros(6, np.int32); lim = np.zeros(32); dt = np.zeros(1)
        if H.kpilqr_host_model_info(self.name, _p(info), _p(lim), _p(dt)) != 0:
            raise ValueError(f"unknown model {name}")
        self.nq, self.nv, self.nu, self.dof, self.dof_quat, self.nr = (int(v) for v in info)
        self.limits = lim[:2 * self.nu].copy(); self.dt = float(dt[0])
        self.tangent = self.dof != self.dof_quat

    def _op(self, op, qpos, qvel, ctrl, other=None, arg=0.0, index=0):
        H = load_host()
        q = np.ascontiguousarray(qpos, np.float64); v = np.ascontiguousarray(qvel, np.float64); u = np.ascontiguousarray(ctrl, np.float64)
        o = None if other is None else np.ascontiguousarray(other, np.float64)
        oq = np.zeros(32); ov = np.zeros(32)
        rc = H.kpilqr_host_model_op(self.name, op, _p(q), _p(v), _p(u), _p(o), float(arg), int(index), _p(oq), _p(ov))
        if rc != 0:
            raise RuntimeError(f"kpilqr_host_model_op({op}) failed: {rc}")
        return oq, ov

    def step(self, qpos, qvel, ctrl):
        oq, ov = self._op(0, qpos, qvel, ctrl)
        return oq[:self.nq].copy(), ov[:self.nv].copy()

    def residuals(self, qpos, qvel, ctrl):
        return self._op(1, qpos, qvel, ctrl)[0][:self.nr].copy()

    def state_vector(self, qpos, qvel, ctrl=None):
        return self._op(2, qpos, qvel, np.zeros(self.nu) if ctrl is None else ctrl)[0][:2 * self.dof].copy()

    def integrate_pos(self, qpos, index, eps):
        return self._op(3, qpos, np.zeros(self.nv), np.zeros(self.nu), arg=eps, index=index)[0][:self.nq].copy()

    def differentiate_pos(self, dt, qpos1, qpos2):
        return self._op(4, qpos1, np.zeros(self.nv), np.zeros(self.nu), other=qpos2, arg=dt)[1][:self.nv].copy()

    def host_fd(self, qpos, qvel, ctrl, cols, central=True, eps=1e-6):
        """Differentiator::DynamicsDerivatives + ::ResidualDerivatives of the PRODUCT at one state."""
        H = load_host()
        n = 2 * self.dof
        cols = np.ascontiguousarray(cols, np.int32)
        cap = 3 * len(cols)
        jc = np.zeros(cap, np.int32); jm = np.zeros(cap, np.uint8)
        xp = np.zeros((cap, n)); xm = np.zeros((cap, n)); xn = np.zeros(n)
        r_x = np.zeros((self.nr, n)); r_u = np.zeros((self.nr, self.nu))
        q = np.ascontiguousarray(qpos, np.float64); v = np.ascontiguousarray(qvel, np.float64); u = np.ascontiguousarray(ctrl, np.float64)
        nj = H.kpilqr_host_model_fd(self.name, _p(q), _p(v), _p(u), len(cols), _p(cols), int(central), float(eps),
                                    _p(jc), _p(jm), _p(xp), _p(xm), _p(xn), _p(r_x), _p(r_u))
        if nj < 0:
            raise RuntimeError("kpilqr_host_model_fd failed")
        return dict(job_col=jc[:nj], job_mode=jm[:nj], xplus=xp[:nj], xminus=xm[:nj], xnom=xn, r_x=r_x, r_u=r_u)


TRACE_FIELDS = ("derivatives", "lambda_in", "backward_passes", "lambda_exit", "lambda_after_backward", "old_cost", "new_cost",
                "best", "accepted", "converged", "lambda_out", "n_alpha")


def optimise(model, T=100, max_iter=8, min_iter=2, options="", u_init=None):
    """iLQR_GPU::Optimise on a stand-in model (needs the GPU).  Returns cost history, controls and the per-iteration trace."""
    H = load_host()
    hist = np.full(max_iter + 2, np.nan); trace = np.full((max_iter, 24), np.nan)
    M = Model(model)
    U = np.zeros((T, M.nu))
    ui = None if u_init is None else np.ascontiguousarray(u_init, np.float64)
    it = H.kpilqr_host_optimise(model.encode(), T, max_iter, min_iter, options.encode(), _p(ui), _p(hist), len(hist), _p(trace), _p(U))
    if it < 0:
        raise RuntimeError(f"kpilqr_host_optimise failed: {it}")
    rows = []
    for w in trace[:it]:
        if np.isnan(w[1]):
            break
        d = {k: w[i] for i, k in enumerate(TRACE_FIELDS)}
        na = int(w[11]) if not np.isnan(w[11]) else 0
        d["rollout_costs"] = w[12:12 + na].copy(); d["predicted"] = w[18:18 + na].copy()
        rows.append(d)
    return dict(iterations=it, cost_history=hist[~np.isnan(hist)], U=U, trace=rows)


def dof_importance(K, dof, sampling_k_interval=1, svd=False, threshold=0.0):
    """iLQR_SVR::LeastImportantDofs on gains K [T][n][m] (the ABI's layout): (sums [dof], indices below the threshold)."""
    H = load_host()
    Kc = np.ascontiguousarray(K, np.float64)
    T, n, m = Kc.shape
    sums = np.zeros(dof); rem = np.zeros(dof, np.int32)
    cnt = H.kpilqr_host_dof_importance(_p(Kc), dof, m, T, int(sampling_k_interval), int(svd), float(threshold), _p(sums), _p(rem))
    return sums, rem[:cnt].copy()
