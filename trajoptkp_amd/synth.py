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
        if dense_residuals:
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
    )


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


def upload(engine, p, keypoints=True):
    """Push a problem dict into an Engine (everything the GPU path needs to run one iteration)."""
    if keypoints:
        engine.set_keypoints_rows(p["kp_rows"])
    engine.upload_fd(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"],
                     job_nom=p["job_nom"], xnom=p["xnom"], eps=p["eps"])
    engine.upload_residuals(p["r"], p["r_x"], p["r_u"], p["w_run"], p["w_term"])
    engine.upload_nominal(p["u_nom"], p["ctrl_lim"])
    engine.sync()
