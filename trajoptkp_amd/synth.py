"""Seeded synthetic iLQR problems with the dimensions and structure of the reference's tasks
(SURVEY.md section 8d).  MuJoCo and the reference's scene XMLs are unavailable, so the key-point
Jacobians come from a random mechanical-system model; everything downstream of them (FD payload,
key-points, residuals, weights, control limits) has the reference's shapes and constants.

numpy only; used by tests/ and bench.py to feed BOTH the CPU oracle and the HIP engine with
identical bytes.  Layout: see trajoptkp_amd/engine.py (column-major matrix per step).
"""
import numpy as np

# (dof, m, nr, dt, ctrl limits, running weights, terminal weights)
TASKS = {
    # TaskConfigs/toys/acrobot.yaml:13-42
    "acrobot": dict(dof=2, m=1, nr=5, dt=0.01, lim=[200.0],
                    w_run=[0, 0, 1e-3, 1e-3, 100], w_term=[100, 100, 1, 1, 100]),
    # TaskConfigs/free_motion/reaching.yaml:12-30
    "panda_reaching": dict(dof=7, m=7, nr=14, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                           w_run=[0.1] * 7 + [0.01] * 7, w_term=[10.0] * 7 + [1.0] * 7),
    # TaskConfigs/rigid_body_manipulation/twoD_push_no_clutter.yaml:11-49
    "panda_pushing": dict(dof=10, m=7, nr=4, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                          w_run=[1.0, 0.5, 0.1, 0.01], w_term=[100.0, 50.0, 1.0, 0.1]),
    # TaskConfigs/rigid_body_manipulation/twoD_push_heavy_clutter.yaml:12-136
    "high_dof_push": dict(dof=31, m=7, nr=11, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                          w_run=[1.0, 0.5] + [0.1] * 9, w_term=[100.0, 50.0] + [1.0] * 9),
    # TaskConfigs/rigid_body_manipulation/twoD_push_light_clutter.yaml: 7 joints + 4 bodies x 3 (n = 38: three tiles)
    "light_clutter_push": dict(dof=19, m=7, nr=7, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                               w_run=[1.0, 0.5] + [0.1] * 5, w_term=[100.0, 50.0] + [1.0] * 5),
    # four state tiles with 1, 2, 3 four-row chunks in the last one (n = 48, 52, 56; high_dof_push has 4): synthetic clutter scenes that
    # exist to instantiate the four-tile kernels' compile-time chunk counts
    "clutter_n48": dict(dof=24, m=7, nr=7, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                        w_run=[1.0, 0.5] + [0.1] * 5, w_term=[100.0, 50.0] + [1.0] * 5),
    "clutter_n52": dict(dof=26, m=7, nr=7, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                        w_run=[1.0, 0.5] + [0.1] * 5, w_term=[100.0, 50.0] + [1.0] * 5),
    "clutter_n56": dict(dof=28, m=7, nr=7, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                        w_run=[1.0, 0.5] + [0.1] * 5, w_term=[100.0, 50.0] + [1.0] * 5),
    # other control dimensions among the reference's task plugins (joint / actuator counts of
    # TaskConfigs/locomotion/walk_plane.yaml, locomotion/hopper.yaml, toys/pentabot.yaml); weights synthetic
    "walker": dict(dof=9, m=6, nr=4, dt=0.005, lim=[1.0] * 6,
                   w_run=[1.0, 0.5, 0.1, 0.01], w_term=[10.0, 5.0, 1.0, 0.1]),
    "hopper": dict(dof=6, m=3, nr=4, dt=0.005, lim=[1.0] * 3,
                   w_run=[1.0, 0.5, 0.1, 0.01], w_term=[10.0, 5.0, 1.0, 0.1]),
    "pentabot": dict(dof=5, m=3, nr=3, dt=0.01, lim=[5.0] * 3,
                     w_run=[1.0, 0.1, 0.01], w_term=[100.0, 1.0, 0.1]),
    # an under-actuated 5-joint arm with 2 motors: a shape with no instantiation of its own (tiled catch-all)
    "arm5x2": dict(dof=5, m=2, nr=3, dt=0.01, lim=[5.0] * 2,
                   w_run=[1.0, 0.1, 0.01], w_term=[100.0, 1.0, 0.1]),
    # TaskConfigs/locomotion/humanoid.yaml:14-15: 21 actuators, 27 DoFs with the free root (n = 54: four state tiles, two
    # control tiles); weights synthetic.  humanoid_fixed: the same without the root (n = 42: three tiles)
    "humanoid": dict(dof=27, m=21, nr=8, dt=0.005, lim=[1.0] * 21,
                     w_run=[1.0, 0.5, 0.5, 0.1, 0.1, 0.01, 0.01, 0.01], w_term=[100.0, 50.0, 50.0, 1.0, 1.0, 0.1, 0.1, 0.1]),
    "humanoid_fixed": dict(dof=21, m=21, nr=6, dt=0.005, lim=[1.0] * 21,
                           w_run=[1.0, 0.5, 0.1, 0.1, 0.01, 0.01], w_term=[100.0, 50.0, 1.0, 1.0, 0.1, 0.1]),
    # a 12-actuator quadruped (9 < num_ctrl <= 16: one control tile, wide backward kernel)
    "quadruped": dict(dof=18, m=12, nr=6, dt=0.005, lim=[1.0] * 12,
                      w_run=[1.0, 0.5, 0.1, 0.1, 0.01, 0.01], w_term=[100.0, 50.0, 1.0, 1.0, 0.1, 0.1]),
    # a fully actuated 8-joint arm: num_ctrl = 8, the largest the padded tiled backward kernel takes
    "arm8": dict(dof=8, m=8, nr=6, dt=0.008, lim=[50.0] * 8,
                 w_run=[1.0, 0.5, 0.1, 0.1, 0.01, 0.01], w_term=[100.0, 50.0, 1.0, 1.0, 0.1, 0.1]),
}


def seed_for(config_id, b):
    return 0x5EED0000 + config_id * 1_000_003 + b


def keypoint_rows_set_interval(dof, T, min_N):
    """KeypointGenerator::GenerateKeyPointsSetInterval (KeyPointGenerator.cpp:319-339), numpy twin
    of the test oracle's set-interval routine (used only to build inputs)."""
    offs = np.zeros(T + 1, np.int32)
    cols = []
    for t in range(T):
        offs[t] = len(cols)
        if (t < T - 1 and t % min_N == 0) or t == T - 1:
            cols.extend(range(dof))
    offs[T] = len(cols)
    return offs, np.asarray(cols, np.int32)


def _dynamics_keypoints(rng, dof, m, dt, kp_times):
    """A_kp = I + dt [[0, I], [S_q, S_v]], B_kp = dt [[0], [M^-1 (first m cols)]]; smooth random
    walk (2 % per key-point) of the mechanical parameters.  Returns column-major stacks
    A [Kp, n, n] (A[k, c, r]), B [Kp, m, n]."""
    n = 2 * dof
    G = rng.standard_normal((dof, dof))
    Mass = np.diag(rng.uniform(0.5, 3.0, dof)) + 0.1 * G @ G.T / dof
    stiff = rng.uniform(0.0, 20.0, dof)
    damp = rng.uniform(0.1, 2.0, dof)
    Kp = len(kp_times)
    A = np.zeros((Kp, n, n))
    B = np.zeros((Kp, m, n))
    for k in range(Kp):
        if k > 0:
            Mass = Mass * (1.0 + 0.02 * rng.standard_normal((dof, dof)))
            Mass = 0.5 * (Mass + Mass.T)
            Mass += np.diag(np.maximum(0.0, 0.5 - np.diag(Mass)))
            stiff = np.abs(stiff * (1.0 + 0.02 * rng.standard_normal(dof)))
            damp = np.abs(damp * (1.0 + 0.02 * rng.standard_normal(dof)))
        Minv = np.linalg.inv(Mass + 2.0 * np.eye(dof))
        Ak = np.eye(n)
        Ak[:dof, dof:] += dt * np.eye(dof)
        Ak[dof:, :dof] += dt * (-Minv @ np.diag(stiff))
        Ak[dof:, dof:] += dt * (-Minv @ np.diag(damp))
        Bk = np.zeros((n, m))
        Bk[dof:, :] = dt * Minv[:, :m]
        A[k] = Ak.T          # column-major: [c, r]
        B[k] = Bk.T
    return A, B


def make_problem(task="panda_reaching", T=3000, batch=1, min_N=5, config_id=2, dense_residuals=False,
                 one_sided_frac=0.0, lam=0.1, eps=1e-6, first_b=0):
    """Builds one batch of `batch` trajectories.  Returns a dict of numpy arrays in C-ABI layout."""
    cfg = TASKS[task]
    dof, m, nr, dt = cfg["dof"], cfg["m"], cfg["nr"], cfg["dt"]
    n = 2 * dof
    lim = np.asarray(cfg["lim"], dtype=np.float64)
    ctrl_lim = np.stack([-lim, lim], axis=1).reshape(-1)           # lo0,hi0,lo1,hi1...
    offs, cols = keypoint_rows_set_interval(dof, T, min_N)
    kp_times = np.nonzero(np.diff(offs))[0].astype(np.int32)
    Kp = len(kp_times)

    job_b, job_t, job_col, job_mode, job_nom = [], [], [], [], []
    xplus, xminus, xnom = [], [], []
    r = np.zeros((batch, T + 1, nr)); r_x = np.zeros((batch, T + 1, nr, n)); r_u = np.zeros((batch, T + 1, nr, m))
    u_nom = np.zeros((batch, T, m))
    A_kp_all, B_kp_all = [], []
    for b in range(batch):
        rng = np.random.default_rng(seed_for(config_id, first_b + b))
        A_kp, B_kp = _dynamics_keypoints(rng, dof, m, dt, kp_times)
        A_kp_all.append(A_kp); B_kp_all.append(B_kp)
        # FD payload: per key-point, columns [ctrl (B), vel (A col i+dof), pos (A col i)] as the
        # reference's Differentiator orders them; x+/- = xnom +/- eps * column
        x0 = rng.standard_normal((Kp, n))
        ncol = 2 * dof + m
        cols_out = np.concatenate([n + np.arange(m), dof + np.arange(dof), np.arange(dof)]).astype(np.int32)
        J = np.concatenate([B_kp, A_kp[:, dof:, :], A_kp[:, :dof, :]], axis=1)     # [Kp, ncol, n] (columns)
        mode = np.zeros((Kp, ncol), np.uint8)
        if one_sided_frac > 0:
            u = rng.uniform(size=(Kp, ncol))
            mode[u < one_sided_frac] = 1
            mode[u < one_sided_frac / 2] = 2
        xp = x0[:, None, :] + eps * J
        xm = x0[:, None, :] - eps * J
        # one-sided jobs difference against the nominal next state
        xp = np.where((mode == 2)[..., None], x0[:, None, :], xp)
        xm = np.where((mode == 1)[..., None], x0[:, None, :], xm)
        job_b.append(np.full(Kp * ncol, b, np.int32))
        job_t.append(np.repeat(kp_times, ncol))
        job_col.append(np.tile(cols_out, Kp))
        job_mode.append(mode.reshape(-1))
        job_nom.append(np.repeat(np.arange(Kp, dtype=np.int32) + b * Kp, ncol))
        xplus.append(xp.reshape(-1, n)); xminus.append(xm.reshape(-1, n)); xnom.append(x0)
        # residuals: N(0, 0.5^2) decaying linearly to 0.05 at T
        scale = np.linspace(0.5, 0.05, T + 1)[:, None]
        r[b] = rng.standard_normal((T + 1, nr)) * scale
        if dense_residuals == "smooth":
            r_x[b], r_u[b] = smooth_residual_jacobians(rng, T, nr, n, m)
        elif dense_residuals:
            r_x[b] = rng.standard_normal((T + 1, nr, n)) * 0.3
            r_u[b] = rng.standard_normal((T + 1, nr, m)) * 0.05
        else:
            # reaching: r = [q - q*, qdot] -> r_x rows of the identity, r_u = 0 (Reaching.cpp:43-54)
            for i in range(nr):
                r_x[b, :, i, i % n] = 1.0
        u_nom[b] = rng.uniform(-0.3, 0.3, (T, m)) * lim[None, :m]
    return dict(
        task=task, dof=dof, n=n, m=m, nr=nr, T=T, batch=batch, dt=dt, eps=eps, lam=lam,
        kp_rows=[(offs, cols)] * batch, kp_times=kp_times,
        job_b=np.concatenate(job_b), job_t=np.concatenate(job_t), job_col=np.concatenate(job_col),
        job_mode=np.concatenate(job_mode), job_nom=np.concatenate(job_nom),
        xplus=np.concatenate(xplus), xminus=np.concatenate(xminus), xnom=np.concatenate(xnom),
        r=r, r_x=r_x, r_u=r_u, w_run=np.asarray(cfg["w_run"], np.float64), w_term=np.asarray(cfg["w_term"], np.float64),
        u_nom=u_nom, ctrl_lim=ctrl_lim, A_kp=np.stack(A_kp_all), B_kp=np.stack(B_kp_all),
        # the ONE residual Jacobian of the task when it has one (reaching: selector rows, r_u = 0 -- Reaching.cpp:43-54): what a
        # host with analytic residuals gives kpilqr_upload_residual_jacobians_const instead of T+1 copies per trajectory
        rx_const=(None if dense_residuals else r_x[0, 0].copy()),
    )


def contact_trajectory(rng, dof, T, dt, n_joints=7):
    """synthetic nominal state trajectory of a pushing task -- a cubic-spline reach of the arm
    joints (plus a small smooth wiggle, as a tracking controller leaves) and a contact event at t in U(800, 2200)
    (scaled to the horizon) that puts a velocity step of 0.2 m/s on the body DoFs, decaying with friction.
    Returns X [T, 2*dof] (positions, then velocities), the input of the key-point generators."""
    tt = np.arange(T) * dt
    tau = tt / max(tt[-1], dt)
    q = np.zeros((T, dof)); v = np.zeros((T, dof))
    nj = min(n_joints, dof)
    q0 = rng.uniform(-1.0, 1.0, nj); q1 = q0 + rng.uniform(-1.2, 1.2, nj)
    sp = 3 * tau ** 2 - 2 * tau ** 3
    dsp = (6 * tau - 6 * tau ** 2) / max(tt[-1], dt)
    amp = rng.uniform(0.0, 0.03, nj); om = rng.uniform(2.0, 9.0, nj); ph = rng.uniform(0, 2 * np.pi, nj)
    q[:, :nj] = q0 + (q1 - q0) * sp[:, None] + amp * np.sin(om * tt[:, None] + ph)
    v[:, :nj] = (q1 - q0) * dsp[:, None] + amp * om * np.cos(om * tt[:, None] + ph)
    if dof > nj:
        tc = int(rng.uniform(800.0, 2200.0) * T / 3000.0)
        tc = min(max(tc, 2), T - 3)
        dirs = rng.uniform(-1.0, 1.0, dof - nj)
        dirs *= 0.2 / max(np.max(np.abs(dirs)), 1e-9)
        after = np.maximum(tt - tt[tc], 0.0)
        on = (np.arange(T) >= tc)[:, None]
        decay = np.exp(-after / 1.5)[:, None]
        v[:, nj:] = on * dirs * decay
        q[:, nj:] = on * dirs * 1.5 * (1.0 - decay)
        # the arm feels the contact too: a small velocity kick on two joints
        for j in rng.choice(nj, size=min(2, nj), replace=False):
            kick = rng.uniform(-0.05, 0.05)
            v[:, j] += on[:, 0] * kick * decay[:, 0]
            q[:, j] += on[:, 0] * kick * 1.5 * (1.0 - decay[:, 0])
    return np.concatenate([q, v], axis=1)


def dynamics_dense_smooth(rng, dof, m, dt, T, n_events=3):
    """Dense (every step) Jacobians of a random mechanical system whose parameters vary smoothly in time, with a few
    contact-like events that switch the stiffness/damping of single DoFs -- the input of the emulated
    `iterative_error` key-point placement.  Column-major stacks A [T, n, n], B [T, m, n]."""
    n = 2 * dof
    G = rng.standard_normal((dof, dof))
    Mass = np.diag(rng.uniform(0.5, 3.0, dof)) + 0.1 * G @ G.T / dof + 2.0 * np.eye(dof)
    Minv = np.linalg.inv(Mass)
    tt = np.arange(T) * dt
    stiff0 = rng.uniform(0.0, 20.0, dof); damp0 = rng.uniform(0.1, 2.0, dof)
    om = rng.uniform(0.05, 0.6, (2, dof)); ph = rng.uniform(0, 2 * np.pi, (2, dof)); amp = rng.uniform(0.0, 0.5, (2, dof))
    amp[:, rng.uniform(size=dof) < 0.4] = 0.0                          # some DoFs hardly change: few key-points
    stiff = stiff0 * (1.0 + amp[0] * np.sin(om[0] * tt[:, None] + ph[0]))
    damp = damp0 * (1.0 + amp[1] * np.sin(om[1] * tt[:, None] + ph[1]))
    # contact-rich DoFs: band-limited jitter (correlation time ~25 steps) on top of the slow variation
    rough = rng.uniform(size=dof) < 0.5
    kern = np.hanning(51); kern /= kern.sum()
    for j in np.nonzero(rough)[0]:
        stiff[:, j] *= 1.0 + 0.3 * np.convolve(rng.standard_normal(T + 50), kern, mode="valid")
        damp[:, j] *= 1.0 + 0.3 * np.convolve(rng.standard_normal(T + 50), kern, mode="valid")
    for _ in range(n_events):
        te = int(rng.integers(T // 10, T - T // 10)); j = int(rng.integers(0, dof))
        ramp = np.clip((np.arange(T) - te) / 40.0, 0.0, 1.0)
        stiff[:, j] += ramp * rng.uniform(5.0, 30.0)
        damp[:, j] += ramp * rng.uniform(0.5, 3.0)
    A = np.zeros((T, n, n)); B = np.zeros((T, m, n))
    Ar = np.broadcast_to(np.eye(n), (T, n, n)).copy()                   # row-major A[t, r, c]
    Ar[:, :dof, dof:] += dt * np.eye(dof)
    Ar[:, dof:, :dof] += dt * (-Minv[None, :, :] * stiff[:, None, :])
    Ar[:, dof:, dof:] += dt * (-Minv[None, :, :] * damp[:, None, :])
    A[:] = np.swapaxes(Ar, 1, 2)
    Bk = np.zeros((n, m)); Bk[dof:, :] = dt * Minv[:, :m]
    B[:] = Bk.T
    return A, B


def rows_from_dof_lists(dof, T, lists):
    """Per-DoF sorted key-point times -> the reference's rows (CSR over time: offs [T+1], cols)."""
    per_t = [[] for _ in range(T)]
    for i in range(dof):
        for t in lists[i]:
            per_t[int(t)].append(i)
    offs = np.zeros(T + 1, np.int32); cols = []
    for t in range(T):
        offs[t] = len(cols); cols.extend(per_t[t])
    offs[T] = len(cols)
    return offs, np.asarray(cols, np.int32)


def smooth_residual_jacobians(rng, T, nr, n, m):
    """Residual Jacobians with the structure of the reference's manipulation tasks (TwoDPushing.cpp:291-352: distances and
    velocities of bodies, a joint velocity, the end-effector's distance to the object -- functions of the STATE that move
    slowly along a trajectory, none of them of the control): r_x(t) = C0 + C1 sin(w t + phi) element-wise with one to three
    periods over the horizon, r_u = 0.  `dense_residuals=True` draws every step's Jacobians independently instead -- a worst case
    for the backward sweeps, whose running inverse of Q_uu then never applies and every step factorises."""
    t = np.arange(T + 1, dtype=np.float64)[:, None, None]
    C0 = rng.standard_normal((1, nr, n)) * 0.3
    C1 = rng.standard_normal((1, nr, n)) * 0.15
    w = 2.0 * np.pi * rng.integers(1, 4, (1, nr, n)) / max(T, 1)
    phi = rng.uniform(0.0, 2.0 * np.pi, (1, nr, n))
    return C0 + C1 * np.sin(w * t + phi), np.zeros((T + 1, nr, m))


def make_ragged_problem(task, T, kp_rows, dyn=None, config_id=3, dense_residuals=True, one_sided_frac=0.0, lam=0.1,
                        eps=1e-6, first_b=0):
    """Like make_problem, but with one key-point row list PER TRAJECTORY (adaptive_jerk / iterative_error style:
    every DoF has its own key-point times) and FD jobs only where the reference would compute them: at step t the
    columns of the DoFs listed in keypoints[t] (control column i for i < num_ctrl, velocity column i+dof, position
    column i; Differentiator.cpp:81-428).  dyn: optional list of dense (A [T,n,n], B [T,m,n]) per trajectory (else a
    random walk over the union of key-point times)."""
    cfg = TASKS[task]
    dof, m, nr, dt = cfg["dof"], cfg["m"], cfg["nr"], cfg["dt"]
    n = 2 * dof
    batch = len(kp_rows)
    lim = np.asarray(cfg["lim"], dtype=np.float64)
    ctrl_lim = np.stack([-lim, lim], axis=1).reshape(-1)
    job_b, job_t, job_col, job_mode, job_nom, xplus, xminus, xnom = [], [], [], [], [], [], [], []
    r = np.zeros((batch, T + 1, nr)); r_x = np.zeros((batch, T + 1, nr, n)); r_u = np.zeros((batch, T + 1, nr, m))
    u_nom = np.zeros((batch, T, m))
    nom_base = 0
    for b in range(batch):
        rng = np.random.default_rng(seed_for(config_id, first_b + b))
        offs, cols = kp_rows[b]
        kp_t = np.nonzero(np.diff(offs))[0].astype(np.int32)
        if dyn is not None:
            A_kp, B_kp = dyn[b][0][kp_t], dyn[b][1][kp_t]
        else:
            A_kp, B_kp = _dynamics_keypoints(rng, dof, m, dt, kp_t)
        x0 = rng.standard_normal((len(kp_t), n))
        for k, t in enumerate(kp_t):
            ds = np.asarray(cols[offs[t]:offs[t + 1]], np.int64)
            cc = np.concatenate([n + ds[ds < m], dof + ds, ds]).astype(np.int32)          # ctrl, vel, pos
            J = np.concatenate([B_kp[k][ds[ds < m]], A_kp[k][dof + ds], A_kp[k][ds]], axis=0)   # [ncol, n] columns
            mode = np.zeros(len(cc), np.uint8)
            if one_sided_frac > 0:
                u = rng.uniform(size=len(cc))
                mode[u < one_sided_frac] = 1
                mode[u < one_sided_frac / 2] = 2
            xp = x0[k][None, :] + eps * J; xm = x0[k][None, :] - eps * J
            xp = np.where((mode == 2)[:, None], x0[k][None, :], xp)
            xm = np.where((mode == 1)[:, None], x0[k][None, :], xm)
            job_b.append(np.full(len(cc), b, np.int32)); job_t.append(np.full(len(cc), t, np.int32))
            job_col.append(cc); job_mode.append(mode); job_nom.append(np.full(len(cc), nom_base + k, np.int32))
            xplus.append(xp); xminus.append(xm)
        xnom.append(x0); nom_base += len(kp_t)
        scale = np.linspace(0.5, 0.05, T + 1)[:, None]
        r[b] = rng.standard_normal((T + 1, nr)) * scale
        if dense_residuals == "smooth":
            r_x[b], r_u[b] = smooth_residual_jacobians(rng, T, nr, n, m)
        elif dense_residuals:
            r_x[b] = rng.standard_normal((T + 1, nr, n)) * 0.3
            r_u[b] = rng.standard_normal((T + 1, nr, m)) * 0.05
        else:
            for i in range(nr):
                r_x[b, :, i, i % n] = 1.0
        u_nom[b] = rng.uniform(-0.3, 0.3, (T, m)) * lim[None, :m]
    return dict(
        task=task, dof=dof, n=n, m=m, nr=nr, T=T, batch=batch, dt=dt, eps=eps, lam=lam,
        kp_rows=list(kp_rows), kp_times=None,
        job_b=np.concatenate(job_b), job_t=np.concatenate(job_t), job_col=np.concatenate(job_col),
        job_mode=np.concatenate(job_mode), job_nom=np.concatenate(job_nom),
        xplus=np.concatenate(xplus), xminus=np.concatenate(xminus), xnom=np.concatenate(xnom),
        r=r, r_x=r_x, r_u=r_u, w_run=np.asarray(cfg["w_run"], np.float64), w_term=np.asarray(cfg["w_term"], np.float64),
        u_nom=u_nom, ctrl_lim=ctrl_lim, rx_const=(None if dense_residuals else r_x[0, 0].copy()),
    )


def bisect_keypoints(rng, dof, T, min_N, density):
    """Cheap stand-in for the iterative_error placement when no dense A is at hand: recursive bisection of [0, T-1]
    per DoF, splitting an interval with a probability that depends on the DoF (`density` [dof] in 0..1) -- the same
    list structure GenerateKeyPointsIteratively emits (KeyPointGenerator.cpp:449-548): per DoF sorted, first 0, last
    T-1, gaps that are powers-of-two fractions of the horizon."""
    lists = []
    for i in range(dof):
        pts = {0, T - 1}
        stack = [(0, T - 1)]
        while stack:
            a, b = stack.pop()
            if b - a <= min_N:
                continue
            mid = (a + b) // 2
            if rng.uniform() < density[i]:
                pts.add(mid)
                stack.append((a, mid)); stack.append((mid, b))
        lists.append(sorted(pts))
    return rows_from_dof_lists(dof, T, lists)


def tile_problem(p, reps):
    """Replicate a problem `reps` times along the batch axis (bench: few unique seeds, full batch)."""
    B0 = p["batch"]
    q = dict(p)
    q["batch"] = B0 * reps
    nj = len(p["job_b"])
    q["job_b"] = (np.tile(p["job_b"], reps) + np.repeat(np.arange(reps, dtype=np.int32) * B0, nj)).astype(np.int32)
    for k in ("job_t", "job_col", "job_mode"):
        q[k] = np.tile(p[k], reps)
    nn = len(p["xnom"])
    q["job_nom"] = (np.tile(p["job_nom"], reps) + np.repeat(np.arange(reps, dtype=np.int32) * nn, nj)).astype(np.int32)
    for k in ("xplus", "xminus", "xnom"):
        q[k] = np.tile(p[k], (reps, 1))
    for k in ("r", "r_x", "r_u", "u_nom"):
        q[k] = np.tile(p[k], (reps,) + (1,) * (p[k].ndim - 1))
    q["kp_rows"] = p["kp_rows"] * reps
    return q


def kp_ordered_payload(p):
    """The FD job lists of a problem re-laid-out BY its key-point lists -- the payload of kpilqr_upload_fd_kp
    (include/kpilqr.h): xplus / xminus [entries][3][n] and mode [entries], entry = position in the per-DoF CSR, kind 0 / 1 / 2
    = position / velocity / control column of the entry's DoF.  A one-sided job carries the nominal next state in the slot
    of the side that was not perturbed, and its bit in the entry's mode byte.  Jobs at steps that are not key-points of
    their DoF have no slot and are dropped; slots no job fills stay zero."""
    from .engine import rows_to_dof_csr
    dof, n, m, T, B = p["dof"], p["n"], p["m"], p["T"], p["batch"]
    offs, times = rows_to_dof_csr(p["kp_rows"], dof, T)
    E = int(offs[-1])
    entry_of = np.full((B * dof, T), -1, np.int64)
    lists = np.repeat(np.arange(B * dof), np.diff(offs))
    entry_of[lists, times] = np.arange(E)
    col = p["job_col"].astype(np.int64)
    kind = np.where(col < dof, 0, np.where(col < n, 1, 2))
    d = np.where(kind == 0, col, np.where(kind == 1, col - dof, col - n))
    e = entry_of[p["job_b"].astype(np.int64) * dof + d, p["job_t"]]
    ok = e >= 0
    e, kind, jm = e[ok], kind[ok], p["job_mode"][ok]
    xp = np.zeros((E, 3, n)); xm = np.zeros((E, 3, n)); mode = np.zeros(E, np.uint8)
    xnom_rows = p["xnom"][p["job_nom"][ok]] if len(p["xnom"]) else np.zeros((len(e), n))
    xp[e, kind] = np.where((jm == 2)[:, None], xnom_rows, p["xplus"][ok])
    xm[e, kind] = np.where((jm == 1)[:, None], xnom_rows, p["xminus"][ok])
    np.bitwise_or.at(mode, e, ((jm != 0).astype(np.uint8) << kind.astype(np.uint8)))
    return xp, xm, mode


def upload(engine, p, keypoints=True, kp_ordered=False, rx_const=False):
    """Push a problem dict into an Engine (everything the GPU path needs to run one iteration).  kp_ordered: the FD
    payload goes up key-point ordered (kpilqr_upload_fd_kp) instead of as job lists.  rx_const: a problem whose residual
    Jacobian is one constant matrix (p["rx_const"]) uploads THAT, once (kpilqr_upload_residual_jacobians_const), instead of
    the per-step copies."""
    if keypoints:
        engine.set_keypoints_rows(p["kp_rows"])
    if kp_ordered:
        # a large resident payload goes up once from pageable memory (the call waits for it); small ones through a pinned slab
        big = p["xplus"].nbytes > (256 << 20)
        engine.upload_fd_kp(engine.fd_kp_slab(*kp_ordered_payload(p), pinned=not big), eps=p["eps"])
    else:
        engine.upload_fd(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"],
                         job_nom=p["job_nom"], xnom=p["xnom"], eps=p["eps"])
    # a task without control residuals never uploads r_u (the context's buffer starts zeroed): include/kpilqr.h
    if rx_const and p.get("rx_const") is not None:
        engine.upload_residuals(p["r"], None, None, p["w_run"], p["w_term"])
        engine.upload_residual_jacobians_const(p["rx_const"], None)
    else:
        engine.upload_residuals(p["r"], p["r_x"], p["r_u"] if np.any(p["r_u"]) else None, p["w_run"], p["w_term"])
    engine.upload_nominal(p["u_nom"], p["ctrl_lim"])
    engine.sync()
