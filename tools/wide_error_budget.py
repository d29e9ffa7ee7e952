"""Where do the digits of the wide-control family's predicted costs go?  (GPU box)   python tools/wide_error_budget.py [cases] [seed]

Round-4 verdict, Weak 3: in the 4 000-case fuzz sweep `mfma_f64_wide / mfma_f64_wide` (quadruped, humanoid: 8 < num_ctrl <= 32)
reaches a predicted-cost error of 1.9e-10 against the CPU oracle where the test bar is 1e-9 -- the thinnest margin in the tree.
This tool re-runs the wide-control cases of that sweep (tests/_fuzz.py, same seed), takes the case with the worst cost error and
sets BOTH implementations against a third one in extended precision (numpy longdouble, 64-bit mantissa: the arithmetic of
oracle/crosscheck.py's np_backward / np_forward):

    err(oracle  vs extended)    what the problem's conditioning does to ANY float64 implementation
    err(GPU     vs extended)
    forward pass in extended precision on the GPU's OWN gains   -> the part of the cost error that is inherited from K, k
    forward pass in extended precision on the oracle's gains    (the same for the oracle)

If the oracle is as far from the extended-precision answer as the GPU is, the 1.9e-10 is the distance between two float64
roundings of an ill-conditioned recursion, not digits lost by k_forward_tiled_wide's scoring."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(".", "tests"))
import _fuzz
from oracle import crosscheck as cc
from oracle import pipeline
from oracle import oracle as orc
from trajoptkp_amd import Engine, synth

LD = np.longdouble


def ld_solve(M, R):
    """M^-1 R, Gaussian elimination with partial pivoting in longdouble."""
    M = M.astype(LD).copy(); R = R.astype(LD).copy()
    m = M.shape[0]
    for j in range(m):
        p = j + int(np.argmax(np.abs(M[j:, j])))
        if p != j:
            M[[j, p]] = M[[p, j]]; R[[j, p]] = R[[p, j]]
        piv = M[j, j]
        f = M[j + 1:, j] / piv
        M[j + 1:] -= f[:, None] * M[j][None, :]
        R[j + 1:] -= f[:, None] * R[j][None, :]
    for j in range(m - 1, -1, -1):
        R[j] = (R[j] - M[j, j + 1:] @ R[j + 1:]) / M[j, j]
    return R


def ld_backward(A, B, l_x, l_xx, l_u, l_uu, lam):
    A, B, l_x, l_xx, l_u, l_uu = (x.astype(LD) for x in (A, B, l_x, l_xx, l_u, l_uu))
    T, n, m = A.shape[0], A.shape[1], B.shape[2]
    K = np.zeros((T, m, n), LD); k = np.zeros((T, m), LD)
    Vx = l_x[T - 1].copy(); Vxx = l_xx[T - 1].copy()
    dJ = LD(0)
    I = np.eye(m, dtype=LD)
    for t in range(T - 1, -1, -1):
        Qx = l_x[t] + A[t].T @ Vx
        Qu = l_u[t] + B[t].T @ Vx
        Qxx = l_xx[t] + A[t].T @ Vxx @ A[t]
        Quu = l_uu[t] + B[t].T @ Vxx @ B[t]
        Qux = B[t].T @ Vxx @ A[t]
        inv = ld_solve(Quu + LD(lam) * I, I)
        k[t] = -inv @ Qu
        K[t] = -inv @ Qux
        Vx = Qx + K[t].T @ (Quu @ k[t]) + K[t].T @ Qu + Qux.T @ k[t]
        Vxx = Qxx + K[t].T @ (Quu @ K[t]) + K[t].T @ Qux + Qux.T @ K[t]
        Vxx = (Vxx + Vxx.T) / 2
        dJ += k[t] @ Qu + k[t] @ Quu @ k[t]
    return K, k, dJ


def ld_forward(A, B, K, k, l_x, l_xx, l_u, l_uu, u_nom, ctrl_lim, alphas):
    A, B, K, k, l_x, l_xx, l_u, l_uu, u_nom = (np.asarray(x).astype(LD) for x in (A, B, K, k, l_x, l_xx, l_u, l_uu, u_nom))
    T, n, m = A.shape[0], A.shape[1], B.shape[2]
    lo, hi = ctrl_lim[0::2].astype(LD), ctrl_lim[1::2].astype(LD)
    cost = np.zeros(len(alphas), LD); U = np.zeros((len(alphas), T, m), LD)
    for a, alpha in enumerate(alphas):
        dx = np.zeros(n, LD)
        for t in range(T):
            u = np.clip(u_nom[t] + LD(alpha) * k[t] + K[t] @ dx, lo, hi)
            du = u - u_nom[t]
            U[a, t] = u
            cost[a] += l_x[t] @ dx + LD(0.5) * (dx @ l_xx[t] @ dx) + l_u[t] @ du + LD(0.5) * (du @ l_uu[t] @ du)
            dx = A[t] @ dx + B[t] @ du
    return cost, U


def rel(a, b):
    a = np.asarray(a, LD); b = np.asarray(b, LD)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def gpu_case(c):
    """tests/_fuzz.run_case's GPU half, returning the arrays."""
    for key in _fuzz.ENV_KEYS:
        os.environ.pop(key, None)
    T, batch, lam, pd = c["T"], c["batch"], c["lam"], c["pd"]
    if c["ragged"]:
        p = synth.make_ragged_problem(c["task"], T, c["rows"], config_id=c["config_id"], dense_residuals=c["dense_res"], one_sided_frac=c["osf"], lam=lam)
    else:
        p = synth.make_problem(task=c["task"], T=T, batch=batch, min_N=c["min_N"], dense_residuals=c["dense_res"], one_sided_frac=c["osf"], lam=lam, config_id=c["config_id"])
    with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=c["fused"]) as e:
        synth.upload(e, p, kp_ordered=False)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, dJ = e.backward(lam, pd)
        K, k = e.gains()
        cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        var = e.last_launch("backward") + " / " + e.last_launch("forward")
    return p, st, dJ, K, k, cost, U, var


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 23)
    t0 = time.time()
    worst = None
    nwide = 0
    for case in range(N):
        c = _fuzz.draw_case(rng, case)              # (consumes the generator exactly as the sweep does)
        if c["task"] not in ("quadruped", "humanoid_fixed"):
            continue
        p, st, dJ, K, k, cost, U, var = gpu_case(c)
        if "wide" not in var:
            continue
        nwide += 1
        for b in range(c["batch"]):
            o = pipeline.run_trajectory(p, b, lam=c["lam"], pd_stride=c["pd"], want_U=True)
            if o["status"] != 0 or st[b] != 0:
                continue
            e_cost = _fuzz.relerr(cost[b], o["cost_pred"])
            if worst is None or e_cost > worst[0]:
                worst = (e_cost, c, b)
    e_cost, c, b = worst
    print(f"{nwide} wide-control cases of {N} in {time.time() - t0:.0f} s; worst predicted-cost error vs the oracle: {e_cost:.2e}")
    print(f"  case {c['case']}: {c['task']} T={c['T']} B={c['batch']} trajectory {b}, lambda={c['lam']:.3e}, pd_stride={c['pd']}, "
          f"{'per-DoF lists' if c['ragged'] else 'key-points every %d' % c['min_N']}, dense residual Jacobians {c['dense_res']}")
    p, st, dJ, K, k, cost, U, var = gpu_case(c)
    print(f"  kernels: {var}")
    o = pipeline.run_trajectory(p, b, lam=c["lam"], pd_stride=c["pd"], want_U=True)
    A, B = cc._T(o["A"]), cc._T(o["B"])                       # maths layout [t][row][col]; the stages a2-a6 are bit-exact on the GPU
    l_x, l_xx, l_u, l_uu = o["l_x"], cc._T(o["l_xx"]), o["l_u"], cc._T(o["l_uu"])
    alphas = orc.alphas(6)
    Kx, kx, dJx = ld_backward(A, B, l_x, l_xx, l_u, l_uu, c["lam"])
    costx, Ux = ld_forward(A, B, Kx, kx, l_x, l_xx, l_u, l_uu, p["u_nom"][b], p["ctrl_lim"], alphas)
    Kg, Ko = cc._T(K[b]), cc._T(o["K"])
    print("  against the extended-precision (longdouble) recursion, relative to the largest entry:")
    print(f"    gains K          oracle {rel(Ko, Kx):.2e}   GPU {rel(Kg, Kx):.2e}   (GPU vs oracle {rel(Kg, Ko):.2e})")
    print(f"    gains k          oracle {rel(o['k'], kx):.2e}   GPU {rel(k[b], kx):.2e}")
    print(f"    delta_J          oracle {abs(LD(o['delta_J']) - dJx) / abs(dJx):.2e}   GPU {abs(LD(dJ[b]) - dJx) / abs(dJx):.2e}")
    print(f"    predicted costs  oracle {rel(o['cost_pred'], costx):.2e}   GPU {rel(cost[b], costx):.2e}   (GPU vs oracle {rel(cost[b], o['cost_pred']):.2e})")
    print(f"    controls U       oracle {rel(o['U_alpha'], Ux):.2e}   GPU {rel(U[b], Ux):.2e}")
    # the forward pass alone, in extended precision, on each implementation's OWN float64 gains: what the gains' rounding costs
    cg, Ug = ld_forward(A, B, Kg, k[b], l_x, l_xx, l_u, l_uu, p["u_nom"][b], p["ctrl_lim"], alphas)
    co, Uo = ld_forward(A, B, Ko, o["k"], l_x, l_xx, l_u, l_uu, p["u_nom"][b], p["ctrl_lim"], alphas)
    print("  forward pass in extended precision on each side's own float64 gains (the error its scoring did NOT make):")
    print(f"    predicted costs  oracle's gains {rel(co, costx):.2e}   GPU's gains {rel(cg, costx):.2e}")
    print(f"    scoring alone    oracle {rel(o['cost_pred'], co):.2e}   GPU {rel(cost[b], cg):.2e}      (float64 forward pass vs the extended one on the SAME gains)")
    growth = float(np.max(np.abs(Ux.astype(np.float64) - p["u_nom"][b][None])))
    print(f"    largest |dU| along the rollouts {growth:.3g}; condition numbers of Quu + lambda I are not tracked here")


if __name__ == "__main__":
    main()
