"""numpy restatement of the reference's HOST finite-difference loops -- TEST INFRASTRUCTURE ONLY (see
oracle/kpilqr_oracle.h): rows a1 / a5 of SURVEY.md section 8, i.e. what stays on the host around the simulator.

    dynamics_derivatives   Differentiator::DynamicsDerivatives, src/Differentiator/Differentiator.cpp:8-462
    residual_derivatives   Differentiator::ResidualDerivatives, src/Differentiator/Differentiator.cpp:464-663

Both are written on the primitives a simulator offers (step, residuals, state vector, mj_integratePos,
mj_differentiatePos), passed in as a `model` object (trajoptkp_amd.host.Model wraps the stand-in simulators of this
tree, which play MuJoCo's role for both sides of the comparison).  The product's Differentiator emits JOBS for the GPU's
fd_difference stage instead of A and B; the tests difference those jobs with orc_fd_difference and compare with the
matrices formed here, the way the reference forms them.  Parity unpinned beyond the reference's own Derivatives.humanoid
test (src/tests/Derivs_Test.cpp:170-197), which needs MuJoCo.
"""
import numpy as np


def _sv(model, q, v):
    return model.state_vector(q, v)


def dynamics_derivatives(model, qpos, qvel, ctrl, cols, central_diff=True, eps=1e-6):
    """-> A [n, n], B [n, m] (only the columns of the DoFs in `cols` are written, the rest stay 0) -- :441-457."""
    dof, m = model.dof, model.nu
    n = 2 * dof
    lim = model.limits
    q0, v0, u0 = np.array(qpos, float), np.array(qvel, float), np.array(ctrl, float)
    dctrl = np.zeros((n, m)); dqvel = np.zeros((n, dof)); dqpos = np.zeros((n, dof))
    qidx = list(range(dof))                                       # StateIndexToQposIndex: identity for the stand-ins
    qn, vn = model.step(q0, v0, u0)                               # unperturbed next state (:66-71)
    next_state = _sv(model, qn, vn)

    def column(out, i, plus, minus, mode):
        """mode 'c' central, 'f' forward, 'b' backward: position rows by mj_differentiatePos on the full states,
        velocity rows by plain differences of the state vectors (:166-222 and its two repeats)."""
        if mode == "c":
            vel_diff = model.differentiate_pos(2 * eps, minus[0], plus[0])
            sp, sm = _sv(model, *plus), _sv(model, *minus)
            out[:dof, i] = vel_diff[qidx]
            out[dof:, i] = (sp[dof:] - sm[dof:]) / (2 * eps)
        elif mode == "f":
            vel_diff = model.differentiate_pos(eps, qn, plus[0])
            sp = _sv(model, *plus)
            out[:dof, i] = vel_diff[qidx]
            out[dof:, i] = (sp[dof:] - next_state[dof:]) / (eps)
        else:
            vel_diff = model.differentiate_pos(eps, minus[0], qn)
            sm = _sv(model, *minus)
            out[:dof, i] = vel_diff[qidx]
            out[dof:, i] = (next_state[dof:] - sm[dof:]) / (eps)

    for i in range(m):                                            # ---- controls (:81-223)
        if i not in cols:
            continue
        up = u0.copy(); up[i] += eps
        nudge_forward = not (up[i] > lim[2 * i + 1])
        plus = model.step(q0, v0, up) if nudge_forward else None
        um = u0.copy(); um[i] -= eps
        nudge_back = (central_diff or not nudge_forward) and not (um[i] < lim[2 * i])
        minus = model.step(q0, v0, um) if nudge_back else None
        if nudge_forward and nudge_back:
            column(dctrl, i, plus, minus, "c")
        elif nudge_forward:
            column(dctrl, i, plus, None, "f")
        elif nudge_back:
            column(dctrl, i, None, minus, "b")
    for i in range(dof):                                          # ---- velocities (:226-325)
        if i not in cols:
            continue
        vp = v0.copy(); vp[i] += eps
        plus = model.step(q0, vp, u0)
        if central_diff:
            vm = v0.copy(); vm[i] -= eps
            column(dqvel, i, plus, model.step(q0, vm, u0), "c")
        else:
            column(dqvel, i, plus, None, "f")
    for i in range(dof):                                          # ---- positions (:328-428)
        if i not in cols:
            continue
        plus = model.step(model.integrate_pos(q0, qidx[i], eps), v0, u0)
        if central_diff:
            column(dqpos, i, plus, model.step(model.integrate_pos(q0, qidx[i], -eps), v0, u0), "c")
        else:
            column(dqpos, i, plus, None, "f")
    A = np.zeros((n, n)); B = np.zeros((n, m))
    for col in cols:                                              # :441-457
        A[:, col] = dqpos[:, col]
        A[:, col + dof] = dqvel[:, col]
        if col < m:
            B[:, col] = dctrl[:, col]
    return A, B


def residual_derivatives(model, qpos, qvel, ctrl, central_diff=True, eps=1e-6):
    """-> r_x [nr, n], r_u [nr, m] (:464-663)."""
    dof, m, nr = model.dof, model.nu, model.nr
    n = 2 * dof
    lim = model.limits
    q0, v0, u0 = np.array(qpos, float), np.array(qvel, float), np.array(ctrl, float)
    r0 = model.residuals(q0, v0, u0)
    r_x = np.zeros((nr, n)); r_u = np.zeros((nr, m))
    for i in range(m):                                            # :496-556
        up = u0.copy(); up[i] += eps
        nudge_forward = not (up[i] > lim[2 * i + 1])
        r_inc = model.residuals(q0, v0, up) if nudge_forward else None
        um = u0.copy(); um[i] -= eps
        nudge_back = (central_diff or not nudge_forward) and not (um[i] < lim[2 * i])
        r_dec = model.residuals(q0, v0, um) if nudge_back else None
        if nudge_forward and nudge_back:
            r_u[:, i] = (r_inc - r_dec) / (2 * eps)
        elif nudge_forward:
            r_u[:, i] = (r_inc - r0) / (eps)
        elif nudge_back:
            r_u[:, i] = (r0 - r_dec) / (eps)
    for i in range(dof):                                          # :575-623
        vp = v0.copy(); vp[i] += eps
        vm = v0.copy(); vm[i] -= eps
        r_x[:, i + dof] = (model.residuals(q0, vp, u0) - model.residuals(q0, vm, u0)) / (2 * eps)
    for i in range(dof):                                          # :626-656
        r_x[:, i] = (model.residuals(model.integrate_pos(q0, i, eps), v0, u0)
                     - model.residuals(model.integrate_pos(q0, i, -eps), v0, u0)) / (2 * eps)
    return r_x, r_u
