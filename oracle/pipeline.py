"""Runs the CPU oracle over one synthetic problem (trajoptkp_amd.synth.make_problem) stage by stage.
TEST INFRASTRUCTURE ONLY (see oracle/kpilqr_oracle.h)."""
import numpy as np

from . import oracle as orc


def run_trajectory(p, b, lam=None, pd_stride=100, n_alpha=6, want_U=False, stages=("fd", "interp", "cost", "bwd", "fwd")):
    """Oracle results for trajectory b of problem p, as a dict (column-major-per-step numpy arrays)."""
    n, m, nr, T, dof = p["n"], p["m"], p["nr"], p["T"], p["dof"]
    lam = p["lam"] if lam is None else lam
    sel = p["job_b"] == b
    A = np.zeros((T, n, n)); B = np.zeros((T, m, n))
    out = {}
    if "fd" in stages:
        # xnom rows are global; remap to the selected subset
        orc.fd_difference(n, m, p["job_t"][sel], p["job_col"][sel], p["job_mode"][sel],
                          p["job_nom"][sel], p["xplus"][sel], p["xminus"][sel], p["xnom"], p["eps"], A, B)
        out["A_kp"], out["B_kp"] = A.copy(), B.copy()
    if "interp" in stages:
        offs, cols = p["kp_rows"][b]
        orc.interpolate(dof, m, T, offs, cols, A, B)
    out["A"], out["B"] = A, B
    if "cost" in stages:
        l_x, l_xx, l_u, l_uu = orc.cost_derivs(n, m, nr, T, p["r"][b], p["r_x"][b], p["r_u"][b], p["w_run"], p["w_term"])
        out.update(l_x=l_x, l_xx=l_xx, l_u=l_u, l_uu=l_uu)
    if "bwd" in stages:
        st, K, k, dJ = orc.backward(n, m, T, A, B, out["l_x"], out["l_xx"], out["l_u"], out["l_uu"], lam, pd_stride)
        out.update(status=st, K=K, k=k, delta_J=dJ)
    if "fwd" in stages and out.get("status", 1) == 0:
        al = orc.alphas(n_alpha)
        res = orc.forward_linear(n, m, T, al, A, B, out["K"], out["k"], out["l_x"], out["l_xx"], out["l_u"],
                                 out["l_uu"], p["u_nom"][b], p["ctrl_lim"], want_U=want_U)
        if want_U:
            out["cost_pred"], out["U_alpha"] = res
        else:
            out["cost_pred"] = res
        out["alphas"] = al
    return out
