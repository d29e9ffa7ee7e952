"""Independent numpy/scipy float64 restatement of the hot path, written from the maths (not from
oracle/kpilqr_oracle.c) to cross-check the C oracle, and the generator of tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  The reference has no Python form and cannot be built here (SURVEY.md
section 8c), so these golden vectors are NOT reference outputs: they pin the C oracle and the HIP
engine to each other and to this independent implementation.  Stages a6-a9 therefore remain
"parity unpinned" with respect to the reference itself.

Usage:  python -m oracle.crosscheck            # verify C oracle vs numpy on the golden configs
        python -m oracle.crosscheck --write    # (re)generate tests/golden/*.npz
"""
import argparse
import os
import sys

import numpy as np
import scipy.linalg

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

GOLDEN_DIR = os.path.join(_ROOT, "tests", "golden")


# ---- numpy restatement (row-major "maths" arrays: M[t, r, c]) --------------------------------
def np_fd(p, b):
    n, m, T = p["n"], p["m"], p["T"]
    A = np.zeros((T, n, n)); B = np.zeros((T, n, m))
    sel = np.nonzero(p["job_b"] == b)[0]
    for j in sel:
        mode = p["job_mode"][j]
        if mode == 0:
            col = (p["xplus"][j] - p["xminus"][j]) / (2 * p["eps"])
        elif mode == 1:
            col = (p["xplus"][j] - p["xnom"][p["job_nom"][j]]) / p["eps"]
        else:
            col = (p["xnom"][p["job_nom"][j]] - p["xminus"][j]) / p["eps"]
        c = p["job_col"][j]
        if c < n:
            A[p["job_t"][j], :, c] = col
        else:
            B[p["job_t"][j], :, c - n] = col
    return A, B


def np_interp(p, b, A, B):
    """Per DoF: linear interpolation of its column pair (and B column) between its key-points."""
    dof, m, T = p["dof"], p["m"], p["T"]
    offs, cols = p["kp_rows"][b]
    t_of = np.repeat(np.arange(T), np.diff(offs))
    A = A.copy(); B = B.copy()
    for i in range(dof):
        ts = np.unique(t_of[cols == i])
        for s, e in zip(ts[:-1], ts[1:]):
            if e - s < 2:
                continue
            ks = np.arange(s + 1, e)
            f = (ks - s).astype(np.float64)[:, None]
            for c in (i, i + dof):
                add = (A[e, :, c] - A[s, :, c]) / float(e - s)
                A[ks, :, c] = A[s, :, c][None, :] + f * add[None, :]
            if i < m:
                add = (B[e, :, i] - B[s, :, i]) / float(e - s)
                B[ks, :, i] = B[s, :, i][None, :] + f * add[None, :]
    return A, B


def np_cost(p, b):
    T = p["T"]
    r, rx, ru = p["r"][b][:T], p["r_x"][b][:T], p["r_u"][b][:T]
    w = np.repeat(p["w_run"][None, :], T, axis=0)
    w[T - 1] = p["w_term"]
    l_x = np.einsum("ti,ti,tia->ta", 2 * w, r, rx)
    l_xx = np.einsum("ti,tia,tib->tab", 2 * w, rx, rx)
    l_u = np.einsum("ti,ti,tia->ta", 2 * w, r, ru)
    l_uu = np.einsum("ti,tia,tib->tab", 2 * w, ru, ru)
    return l_x, l_xx, l_u, l_uu


def np_backward(A, B, l_x, l_xx, l_u, l_uu, lam, pd_stride=100):
    T, n, m = A.shape[0], A.shape[1], B.shape[2]
    K = np.zeros((T, m, n)); k = np.zeros((T, m))
    Vx = l_x[T - 1].copy(); Vxx = l_xx[T - 1].copy()
    dJ = 0.0; cnt = 0
    for t in range(T - 1, -1, -1):
        cnt += 1
        Qx = l_x[t] + A[t].T @ Vx
        Qu = l_u[t] + B[t].T @ Vx
        Qxx = l_xx[t] + A[t].T @ Vxx @ A[t]
        Quu = l_uu[t] + B[t].T @ Vxx @ B[t]
        Qux = B[t].T @ Vxx @ A[t]
        Qreg = Quu + lam * np.eye(m)
        if cnt >= pd_stride:
            try:
                np.linalg.cholesky(np.tril(Qreg) + np.tril(Qreg, -1).T)
            except np.linalg.LinAlgError:
                return t + 1, K, k, dJ
            cnt = 0
        sym = np.tril(Qreg) + np.tril(Qreg, -1).T          # LDLT reads the lower triangle
        inv = scipy.linalg.solve(sym, np.eye(m), assume_a="sym")
        k[t] = -inv @ Qu
        K[t] = -inv @ Qux
        Vx = Qx + K[t].T @ (Quu @ k[t]) + K[t].T @ Qu + Qux.T @ k[t]
        Vxx = Qxx + K[t].T @ (Quu @ K[t]) + K[t].T @ Qux + Qux.T @ K[t]
        Vxx = (Vxx + Vxx.T) / 2
        dJ += k[t] @ Qu + k[t] @ Quu @ k[t]
    return 0, K, k, dJ


def np_forward(A, B, K, k, l_x, l_xx, l_u, l_uu, u_nom, ctrl_lim, alphas):
    T, n, m = A.shape[0], A.shape[1], B.shape[2]
    lo, hi = ctrl_lim[0::2], ctrl_lim[1::2]
    cost = np.zeros(len(alphas)); U = np.zeros((len(alphas), T, m))
    for a, alpha in enumerate(alphas):
        dx = np.zeros(n)
        for t in range(T):
            u = np.clip(u_nom[t] + alpha * k[t] + K[t] @ dx, lo, hi)
            du = u - u_nom[t]
            U[a, t] = u
            cost[a] += l_x[t] @ dx + 0.5 * dx @ l_xx[t] @ dx + l_u[t] @ du + 0.5 * du @ l_uu[t] @ du
            dx = A[t] @ dx + B[t] @ du
    return cost, U


def np_pipeline(p, b, lam=None, pd_stride=100, n_alpha=6):
    lam = p["lam"] if lam is None else lam
    A, B = np_fd(p, b)
    A, B = np_interp(p, b, A, B)
    l_x, l_xx, l_u, l_uu = np_cost(p, b)
    st, K, k, dJ = np_backward(A, B, l_x, l_xx, l_u, l_uu, lam, pd_stride)
    alphas = (np.arange(1, n_alpha + 1) / n_alpha) ** 2
    cost, U = np_forward(A, B, K, k, l_x, l_xx, l_u, l_uu, p["u_nom"][b], p["ctrl_lim"], alphas)
    return dict(A=A, B=B, l_x=l_x, l_xx=l_xx, l_u=l_u, l_uu=l_uu, status=st, K=K, k=k, delta_J=dJ,
                cost_pred=cost, U_alpha=U)


def _T(x):            # column-major-per-step <-> maths
    return np.swapaxes(x, -1, -2)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# ---- golden configurations ------------------------------------------------------------------------
GOLDEN = {
    # name: make_problem kwargs (+ pd_stride)
    "panda_T64": dict(task="panda_reaching", T=64, batch=2, min_N=5, config_id=2, dense_residuals=True,
                      one_sided_frac=0.15),
    "acrobot_T100": dict(task="acrobot", T=100, batch=1, min_N=5, config_id=1, dense_residuals=True),
    "pushing_T48": dict(task="panda_pushing", T=48, batch=1, min_N=4, config_id=3, dense_residuals=True),
}
# full-size case: only checksums are stored
GOLDEN_BIG = {"panda_T3000": dict(task="panda_reaching", T=3000, batch=1, min_N=5, config_id=2)}


def compare(name, kw, verbose=True):
    from trajoptkp_amd import synth
    from oracle import pipeline
    p = synth.make_problem(**kw)
    worst = {}
    res = []
    for b in range(p["batch"]):
        o = pipeline.run_trajectory(p, b, want_U=True)
        q = np_pipeline(p, b)
        pairs = dict(A=(_T(o["A"]), q["A"]), B=(_T(o["B"]), q["B"]), l_x=(o["l_x"], q["l_x"]),
                     l_xx=(_T(o["l_xx"]), q["l_xx"]), l_u=(o["l_u"], q["l_u"]), l_uu=(_T(o["l_uu"]), q["l_uu"]),
                     K=(_T(o["K"]), q["K"]), k=(o["k"], q["k"]),
                     delta_J=(np.array(o["delta_J"]), np.array(q["delta_J"])),
                     cost_pred=(o["cost_pred"], q["cost_pred"]), U_alpha=(o["U_alpha"], q["U_alpha"]))
        assert o["status"] == q["status"] == 0, (o["status"], q["status"])
        for key, (x, y) in pairs.items():
            worst[key] = max(worst.get(key, 0.0), rel(x, y))
        res.append(o)
    if verbose:
        print(f"[{name}] C oracle vs numpy, max relative difference:",
              ", ".join(f"{k}={v:.1e}" for k, v in worst.items()))
    return p, res, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    ok = True
    for name, kw in {**GOLDEN, **GOLDEN_BIG}.items():
        p, res, worst = compare(name, kw)
        tol = dict(A=1e-12, B=1e-12, l_x=1e-12, l_xx=1e-12, l_u=1e-12, l_uu=1e-12, K=1e-8, k=1e-8,
                   delta_J=1e-8, cost_pred=1e-8, U_alpha=1e-8)
        for key, v in worst.items():
            if not v <= tol[key]:
                ok = False
                print(f"  MISMATCH {name}.{key}: {v:.3e} > {tol[key]:.0e}")
        if args.write:
            out = {"kwargs": np.array(repr(kw))}
            if name in GOLDEN:
                for b, o in enumerate(res):
                    for key in ("A", "B", "l_x", "l_xx", "l_u", "l_uu", "K", "k", "cost_pred"):
                        out[f"b{b}_{key}"] = o[key]
                    out[f"b{b}_delta_J"] = np.array(o["delta_J"])
            else:
                o = res[0]
                for key in ("A", "B", "l_xx", "K", "k", "cost_pred"):
                    out[f"sum_{key}"] = np.array(np.sum(o[key]))
                    out[f"abssum_{key}"] = np.array(np.sum(np.abs(o[key])))
                out["K_first"] = o["K"][0]; out["K_last"] = o["K"][-1]; out["K_mid"] = o["K"][1500]
                out["k_first"] = o["k"][0]
                out["delta_J"] = np.array(o["delta_J"]); out["cost_pred"] = o["cost_pred"]
            np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **out)
            print("  wrote", name + ".npz")
    print("OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
