"""Host-side C++ classes (trajoptkp_amd/host/): key-point generators against the CPU oracle (CPU), and
the Optimiser-shaped iLQR_GPU end to end on the acrobot plumbing configuration (GPU)."""
import numpy as np
import pytest

from oracle import oracle as orc
from trajoptkp_amd import host


def _states(rng, dof, T, dt=0.01):
    t = np.arange(T) * dt
    q = np.stack([rng.uniform(0.2, 1.0) * np.cos(rng.uniform(1, 8) * t + rng.uniform(0, 3)) for _ in range(dof)], 1)
    v = np.stack([rng.uniform(0.5, 3.0) * np.sin(rng.uniform(1, 9) * t + rng.uniform(0, 3)) for _ in range(dof)], 1)
    v += 0.05 * rng.standard_normal(v.shape)
    return np.concatenate([q, v], 1)


@pytest.mark.parametrize("dof,T,min_N", [(2, 100, 2), (2, 100, 3), (7, 3000, 5), (7, 64, 1)])
def test_set_interval_matches_oracle(dof, T, min_N):
    o, c, pct = host.keypoints("set_interval", dof, T, min_N)
    oo, oc = orc.kp_set_interval(dof, T, min_N)
    assert np.array_equal(o, oo) and np.array_equal(c, oc)
    assert np.allclose(pct, orc.kp_percentages(dof, T, oo, oc))


@pytest.mark.parametrize("seed", range(6))
def test_adaptive_jerk_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    dof, T = int(rng.integers(1, 8)), int(rng.integers(20, 300))
    X = _states(rng, dof, T)
    thr = rng.uniform(1.0, 400.0, dof)
    min_N, max_N = int(rng.integers(1, 4)), int(rng.integers(4, 30))
    o, c, _ = host.keypoints("adaptive_jerk", dof, T, min_N, max_N, thr, dt=0.01, X=X)
    oo, oc = orc.kp_adaptive_jerk(dof, T, min_N, max_N, thr, 0.01, X)
    assert np.array_equal(o, oo) and np.array_equal(c, oc)


@pytest.mark.parametrize("seed", range(6))
def test_adaptive_accel_matches_oracle(seed):
    rng = np.random.default_rng(300 + seed)
    dof, T = int(rng.integers(1, 8)), int(rng.integers(20, 300))
    X = _states(rng, dof, T)
    thr = rng.uniform(0.001, 0.05, dof)
    min_N, max_N = int(rng.integers(1, 4)), int(rng.integers(4, 30))
    o, c, _ = host.keypoints("adaptive_accel", dof, T, min_N, max_N, thr, dt=0.01, X=X)
    oo, oc = orc.kp_adaptive_accel(dof, T, min_N, max_N, thr, X)
    assert np.array_equal(o, oo) and np.array_equal(c, oc)
    assert len(oc) > 2 * dof                      # the thresholds bite: more than the two enforced rows


@pytest.mark.parametrize("seed", range(6))
def test_velocity_change_matches_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    dof, T = int(rng.integers(1, 8)), int(rng.integers(20, 300))
    X = _states(rng, dof, T)
    thr = rng.uniform(0.5, 10.0, dof)
    min_N, max_N = int(rng.integers(1, 4)), int(rng.integers(4, 30))
    o, c, _ = host.keypoints("velocity_change", dof, T, min_N, max_N, thr, X=X)
    oo, oc = orc.kp_velocity_change(dof, T, min_N, max_N, thr, X)
    assert np.array_equal(o, oo) and np.array_equal(c, oc)


@pytest.mark.parametrize("seed", range(4))
def test_iterative_error_matches_oracle(seed):
    rng = np.random.default_rng(200 + seed)
    dof, T = int(rng.integers(1, 5)), int(rng.integers(30, 200))
    n = 2 * dof
    A = np.cumsum(0.01 * rng.standard_normal((T, n, n)), axis=0)
    A[T // 3:, :, :] += 0.3 * rng.standard_normal((n, n))            # a kink forces refinement
    thr = 10.0 ** rng.uniform(-5, -2)
    min_N = int(rng.integers(1, 4))
    o, c, pct = host.keypoints("iterative_error", dof, T, min_N, iterative_error_threshold=thr, A=A)
    oo, oc = orc.kp_iterative_error(dof, T, min_N, thr, A)
    assert np.array_equal(o, oo) and np.array_equal(c, oc)
    assert np.allclose(pct, orc.kp_percentages(dof, T, oo, oc))


@pytest.mark.gpu
def test_acrobot_plumbing_optimise_on_gpu():
    """BASELINE configs[0] end to end through the reference-shaped C++ surface: acrobot swing-up, T=100,
    set_interval 5: RolloutTrajectory -> Iteration (host FD -> GPU fd/interp/cost/backward/forward ->
    confirming rollout).  The cost must decrease and never increase across accepted iterations."""
    res = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2)              # task weights (torque weight 100)
    h = res["cost_history"]
    assert res["iterations"] >= 1 and len(h) >= 2
    assert np.all(np.diff(h) <= 1e-12) and h[1] < h[0], h
    assert np.all(np.isfinite(res["U"])) and np.all(np.abs(res["U"]) <= 100.0 + 1e-9)
    # cheap torque: the optimiser must find a large improvement, monotonically
    res = host.run_acrobot(T=100, min_N=5, max_iter=10, min_iter=3, torque_weight=1e-3)
    h = res["cost_history"]
    assert np.all(np.diff(h) <= 1e-12), h
    assert h[-1] < 0.5 * h[0], h
    assert np.all(np.abs(res["U"]) <= 100.0 + 1e-9) and np.max(np.abs(res["U"])) > 0.1
    # key-point methods that depend on the trajectory / interleave FD with placement
    for method in ("iterative_error", "adaptive_jerk", "velocity_change"):
        r2 = host.run_acrobot(T=100, min_N=2, max_iter=3, min_iter=1, method=method, torque_weight=1e-3)
        assert r2["cost_history"][1] < r2["cost_history"][0], (method, r2["cost_history"])


def test_fd_harness_pool_matches_spawn_per_call_and_is_deterministic():
    """the persistent-pool FD harness produces exactly the job set of the reference-shaped
    spawn-per-call path (bit for bit, order-independent checksum) in an order that does not depend on thread
    scheduling.  CPU only (the acrobot stand-in simulator)."""
    a = host.fd_bench(T=400, reps=2, mode=0, fd_threads=8)
    b = host.fd_bench(T=400, reps=2, mode=1, fd_threads=8)
    c = host.fd_bench(T=400, reps=3, mode=1, fd_threads=3)
    assert a["columns"] == b["columns"] == 2 * 400 * 5            # (ctrl + vel + pos) + (vel + pos) per step
    assert a["checksum_set"] == b["checksum_set"] == c["checksum_set"]
    assert b["checksum_order"] == c["checksum_order"]             # in-place slices: order fixed by key-point time
    assert b["pool"] >= 1


@pytest.mark.gpu
def test_acrobot_fused_unfused_and_analytic_residual_jacobians_agree():
    """The optimiser shim on the fused sweeps, on the materialising pipeline (default at batch 1), and with closed-form
    residual Jacobians: same accepted cost sequence to FD accuracy."""
    base = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method="set_interval+fused", torque_weight=1e-3)
    unf = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method="set_interval+unfused", torque_weight=1e-3)
    ana = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method="set_interval+analytic", torque_weight=1e-3)
    assert base["iterations"] == unf["iterations"] == ana["iterations"]
    assert np.allclose(base["cost_history"], unf["cost_history"], rtol=1e-7)
    assert np.allclose(base["cost_history"], ana["cost_history"], rtol=1e-5)
    assert np.allclose(base["K0"], unf["K0"], rtol=1e-6, atol=1e-9)
    # a2 on the host (iLQR_GPU::host_differencing -> kpilqr_upload_kp_columns): the same bytes as differencing on the device
    col = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method="set_interval+fused+columns", torque_weight=1e-3)
    assert col["iterations"] == base["iterations"]
    assert np.array_equal(col["cost_history"], base["cost_history"]) and np.array_equal(col["K0"], base["K0"])
    # a task that DECLARES its residual Jacobians constant (ModelTranslator::ConstantResidualJacobians; reaching is one:
    # Reaching.cpp:43-54): the shim uploads the pair once per context and never runs a5 -- the same matrices as the closed-form
    # per-step path, so the same bytes come back (kpilqr_upload_residual_jacobians_const's contract)
    for spec in ("set_interval+fused", "set_interval+unfused"):
        per = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method=spec + "+analytic", torque_weight=1e-3)
        con = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method=spec + "+analytic+constjac", torque_weight=1e-3)
        assert con["constant_jacobian_uploads"] == 1 and con["per_step_jacobian_uploads"] == 0, con
        assert per["constant_jacobian_uploads"] == 0 and per["per_step_jacobian_uploads"] >= 1, per
        assert con["iterations"] == per["iterations"]
        assert np.array_equal(con["cost_history"], per["cost_history"]) and np.array_equal(con["K0"], per["K0"]) and np.array_equal(con["U"], per["U"])
    # a filtering task (Optimiser::FilterDynamicsMatrices) runs on the materialising pipeline and still optimises
    for f in ("low_pass", "FIR"):
        r = host.run_acrobot(T=100, min_N=5, max_iter=6, min_iter=2, method=f"set_interval+{f}", torque_weight=1e-3)
        assert np.all(np.diff(r["cost_history"]) <= 1e-12) and r["cost_history"][1] < r["cost_history"][0], (f, r["cost_history"])


def test_on_disk_formats_match_the_reference_writers(tmp_path):
    """the CSV dumps are what FileHandler::SaveTrajecInformation / SaveKeypointsToFile / SaveTaskToFile
    and GenTestingData write: row-major matrices, horizon-1 lines, 6 significant digits, a comma after EVERY
    value, int-accumulated timing columns."""
    rng = np.random.default_rng(1)
    T, dof, m = 6, 2, 1
    n = 2 * dof
    A = rng.standard_normal((T, n, n)) * 10.0 ** rng.integers(-8, 8, (T, n, n))       # A[t, col, row]
    B = rng.standard_normal((T, m, n))
    X = rng.standard_normal((T, n)); U = rng.standard_normal((T, m))
    root = str(tmp_path / "savedTrajecInfo" / "acrobot" / "0")
    assert host.save_trajec(root, A, B, X, U) == 0
    g = lambda v: "%g" % v                                    # default-formatted ostream << double

    def lines(name):
        return open(f"{root}/{name}").read().split("\n")
    la = lines("A_matrices.csv")
    assert la[-1] == "" and len(la) - 1 == T - 1                # steps 0 .. horizon-2
    for t in range(T - 1):
        assert la[t] == "".join(g(A[t, k, j]) + "," for j in range(n) for k in range(n))
    lb = lines("B_matrices.csv")
    for t in range(T - 1):
        assert lb[t] == "".join(g(B[t, k, j]) + "," for j in range(n) for k in range(m))
    assert lines("states.csv")[2] == "".join(g(v) + "," for v in X[2])
    assert lines("controls.csv")[T - 2] == "".join(g(v) + "," for v in U[T - 2])
    # key-points: one line per DoF
    offs, cols = orc.kp_set_interval(dof, 12, 5)
    assert host.save_keypoints(root, offs, cols) == 0
    assert lines("keypoints.csv")[:dof] == ["0,5,10,11,"] * dof
    # task rows round trip through the loader; a size mismatch is refused
    task = str(tmp_path / "TestTasks" / "acrobot" / "3.csv")
    start, targets = np.array([3.1415, 0.3]), np.array([0.0, 0.0, 1.5e-7])
    assert host.save_task(task, start, targets) == 0
    assert open(task).read() == "3.1415,0.3,0,0,1.5e-07,\n"
    s, t_ = host.load_task(task, 2, 3)
    assert np.array_equal(s, start) and np.array_equal(t_, targets)
    assert host.load_task(task, 2, 2) is None and host.load_task(task + ".missing", 2, 3) is None
    # summary.csv: header, and the "Average time" columns are int-accumulated totals
    summ = str(tmp_path / "summary.csv")
    rows = np.array([[0.912345678, 41.25, 5, 7.0, 20.5]])
    tim = np.array([[[1.6, 2.7, 0.9], [0.4, 0.4, 0.4], [10.2, 3.9, 0.5]]])
    assert host.save_summary(summ, rows, tim) == 0
    txt = open(summ).read().split("\n")
    assert txt[0] == ("Cost reduction,Optimisation time (ms),Number iterations,Average num dofs,Average percent derivs,"
                      "Average time derivs (ms),Average time BP (ms),Average time FP (ms)")
    assert txt[1] == "0.912346,41.25,5,7,20.5,3,0,13"        # int(int(0+1.6)+2.7)=3 -> 3; 0; int(10.2)=10,13,13


@pytest.mark.gpu
def test_batched_optimiser_matches_single_trajectory_runs():
    """iLQR_GPU_Batch: B acrobot problems from different starts through ONE context (dims.batch = B) follow,
    trajectory by trajectory, the same accepted-cost sequence as B separate single-trajectory optimisations --
    per-trajectory lambda schedule, PD retry, acceptance and convergence included."""
    q0s = np.array([[3.1415, 0.3], [2.6, -0.4], [3.5, 0.1], [1.2, 0.8]])
    for fused in (False, True):
        res = host.run_acrobot_batch(q0s, T=100, min_N=5, max_iter=7, min_iter=2, torque_weight=1e-3, fused=fused)
        for b, q0 in enumerate(q0s):
            single = host.run_acrobot(T=100, min_N=5, max_iter=7, min_iter=2, torque_weight=1e-3,
                                      method=f"set_interval+{'fused' if fused else 'unfused'}+q0={q0[0]},{q0[1]}")
            assert res["iterations"][b] == single["iterations"], (fused, b, res["iterations"], single["iterations"])
            assert np.allclose(res["cost_history"][b], single["cost_history"], rtol=1e-9), (fused, b)
            assert np.allclose(res["U"][b], single["U"], rtol=1e-7, atol=1e-9)
        assert res["stats"][7] >= 1.0 and np.all(np.isfinite(res["stats"]))
    # the task's ONE residual Jacobian pair uploaded once for the whole batch (ModelTranslator::ConstantResidualJacobians)
    res = host.run_acrobot_batch(q0s, T=100, min_N=5, max_iter=7, min_iter=2, torque_weight=1e-3, fused=True, method="set_interval+constjac")
    for b, q0 in enumerate(q0s):
        single = host.run_acrobot(T=100, min_N=5, max_iter=7, min_iter=2, torque_weight=1e-3, method=f"set_interval+fused+analytic+constjac+q0={q0[0]},{q0[1]}")
        assert res["iterations"][b] == single["iterations"] and np.allclose(res["cost_history"][b], single["cost_history"], rtol=1e-9), b


@pytest.mark.gpu
@pytest.mark.parametrize("method,min_N", [("adaptive_jerk", 2), ("velocity_change", 1), ("adaptive_accel", 3)])
def test_batched_optimiser_partial_regeneration_with_moving_keypoint_counts(method, min_N):
    """Fused batch context, adaptive key-point methods: every trajectory has its own per-DoF lists whose counts change with
    every linearisation, and a trajectory whose step was rejected does NOT regenerate (iLQR.cpp:419) while the others do --
    its entries then sit at shifted offsets of the batch CSR.  The batch must still follow, trajectory by trajectory, the
    single-trajectory runs (round-3 advisor finding: the non-regenerating trajectories ran on stale / zero columns)."""
    q0s = np.array([[3.1415, 0.3], [2.6, -0.4], [3.5, 0.1], [1.2, 0.8], [0.4, -1.1], [2.9, 0.9]])
    for fused in (True, False):
        res = host.run_acrobot_batch(q0s, T=120, min_N=min_N, max_iter=9, min_iter=2, torque_weight=1e-3, fused=fused, method=method)
        rejected = 0
        for b, q0 in enumerate(q0s):
            single = host.run_acrobot(T=120, min_N=min_N, max_iter=9, min_iter=2, torque_weight=1e-3,
                                      method=f"{method}+{'fused' if fused else 'unfused'}+q0={q0[0]},{q0[1]}")
            assert res["iterations"][b] == single["iterations"], (method, fused, b, res["iterations"], single["iterations"])
            assert np.allclose(res["cost_history"][b], single["cost_history"], rtol=1e-9), (method, fused, b)
            assert np.allclose(res["U"][b], single["U"], rtol=1e-7, atol=1e-9)
            h = res["cost_history"][b]
            rejected += int(np.count_nonzero(np.diff(h)[:-1] == 0.0))      # a rejected step that was followed by another iteration
        print(f"{method} fused={fused}: iterations {list(res['iterations'])}, rejected-then-continued steps {rejected}")
        if method == "adaptive_accel":        # (measured: one trajectory has a step rejected and carries on while the others regenerate --
            assert rejected >= 1              # the case the relocation exists for; the other two methods never reject on these starts)


def test_relocate_records_moves_every_kept_trajectory_without_overwriting_one():
    """iLQR_GPU_Batch's key-point records when the batch CSR moves under trajectories that do not regenerate (round-4 advisor: the
    forward-then-backward move order was correct by inspection only).  Random CSRs with shrinking AND growing neighbours, in place
    and into a second slab: every kept trajectory's records must arrive byte for byte at its new entry offset.  CPU only."""
    rng = np.random.default_rng(11)
    stride, dof = 24, 3
    for trial in range(200):
        B = int(rng.integers(2, 9))
        regen = rng.integers(0, 2, B).astype(np.uint8)
        if trial % 7 == 0: regen[:] = 0
        old_cnt = rng.integers(0, 9, (B, dof)); new_cnt = old_cnt.copy()
        for b in range(B):
            if regen[b]: new_cnt[b] = rng.integers(0, 12, dof)       # regenerated lists shrink or grow; kept ones keep their length
        old_offs = np.concatenate([[0], np.cumsum(old_cnt.ravel())]); new_offs = np.concatenate([[0], np.cumsum(new_cnt.ravel())])
        total = int(max(old_offs[-1], new_offs[-1])) + 4
        slab = rng.integers(0, 256, total * stride, dtype=np.uint8)
        want = {b: slab[old_offs[b * dof] * stride:old_offs[(b + 1) * dof] * stride].copy() for b in range(B) if not regen[b]}
        for in_place in (True, False):
            out = host.relocate_records(slab.copy(), stride, B, dof, old_offs, new_offs, regen, in_place=in_place)
            for b, rec in want.items():
                w = int(new_offs[b * dof]) * stride
                assert np.array_equal(out[w:w + len(rec)], rec), (trial, in_place, b)


# ---- a1 / a5: the host finite differences against the numpy restatement of the reference's loops ------------------------
def _fd_states(model, rng, k):
    out = []
    for _ in range(k):
        if model.name == b"acrobot":
            q = rng.uniform(-3, 3, 2)
        else:
            ax = rng.standard_normal(3); ax /= np.linalg.norm(ax); th = rng.uniform(0.1, 2.5)
            q = np.concatenate([rng.uniform(-1, 1, 3), [np.cos(th / 2)], np.sin(th / 2) * ax])
        out.append((q, rng.uniform(-1.5, 1.5, model.nv), rng.uniform(-0.9, 0.9, model.nu) * model.limits[1::2]))
    return out


@pytest.mark.parametrize("name", ["acrobot", "floating_body"])
def test_host_finite_differences_match_the_restated_reference_loops(name):
    """Differentiator::DynamicsDerivatives (jobs -> a2 differencing by the oracle) and ::ResidualDerivatives against
    oracle/host_fd.py: hinge model and free-joint model (tangent-space position rows), central and one-sided control
    columns (control at its limit), a subset of DoFs."""
    from oracle import host_fd, oracle as orc
    M = host.Model(name)
    rng = np.random.default_rng(11)
    n, m, dof = 2 * M.dof, M.nu, M.dof
    for case, (q, v, u) in enumerate(_fd_states(M, rng, 4)):
        cols = list(range(dof)) if case % 2 == 0 else sorted(rng.choice(dof, size=max(1, dof // 2), replace=False).tolist())
        if case >= 2:                              # a control at its upper / lower limit: one-sided columns (:94-143)
            u = u.copy(); u[0] = M.limits[1] if case == 2 else M.limits[0]
        g = M.host_fd(q, v, u, cols)
        nj = len(g["job_col"])
        assert nj == sum(2 + (i < m) for i in cols)
        A = np.zeros((1, n, n)); B = np.zeros((1, m, n))
        orc.fd_difference(n, m, np.zeros(nj, np.int32), g["job_col"], g["job_mode"], np.zeros(nj, np.int32),
                          g["xplus"], g["xminus"], g["xnom"][None, :], 1e-6, A, B)
        A_ref, B_ref = host_fd.dynamics_derivatives(M, q, v, u, cols)
        if case >= 2 and 0 in cols:
            assert g["job_mode"][list(g["job_col"]).index(n)] == (2 if case == 2 else 1)
        # column-major per step: A[0, c, r]
        assert np.array_equal(A[0].T, A_ref), (name, case, np.max(np.abs(A[0].T - A_ref)))
        assert np.array_equal(B[0].T, B_ref), (name, case, np.max(np.abs(B[0].T - B_ref)))
        r_x, r_u = host_fd.residual_derivatives(M, q, v, u)
        assert np.array_equal(g["r_x"], r_x) and np.array_equal(g["r_u"], r_u), (name, case)


def test_free_joint_jacobian_is_the_tangent_space_jacobian():
    """The free-joint columns are derivatives in the tangent space: they agree with an independent Jacobian obtained by
    perturbing with rotation vectors on the LEFT-composed chart of the nominal next state (numpy quaternion algebra)."""
    M = host.Model("floating_body")
    rng = np.random.default_rng(3)
    (q, v, u), = _fd_states(M, rng, 1)
    g = M.host_fd(q, v, u, list(range(6)))
    n, eps = 12, 1e-6
    A = np.zeros((n, n))
    for j, col in enumerate(g["job_col"]):
        if col < n:
            A[:, col] = (g["xplus"][j] - g["xminus"][j]) / (2 * eps)

    def qmul(a, b):
        return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])

    def qexp(w):
        th = np.linalg.norm(w)
        return np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * w / th]) if th > 0 else np.array([1.0, 0, 0, 0])

    def qlog(qq):
        qq = qq if qq[0] >= 0 else -qq
        s = np.linalg.norm(qq[1:])
        return 2 * np.arctan2(s, qq[0]) * qq[1:] / s if s > 0 else np.zeros(3)

    qn, vn = M.step(q, v, u)

    def f(dx):          # tangent perturbation of (q, v) -> tangent difference of the next state from the nominal one
        qq = np.concatenate([q[:3] + dx[:3], qmul(q[3:], qexp(dx[3:6]))])
        q2, v2 = M.step(qq, v + dx[6:], u)
        conj = q[3:].copy()
        conj = qn[3:] * np.array([1, -1, -1, -1])
        return np.concatenate([q2[:3] - qn[:3], qlog(qmul(conj, q2[3:])), v2 - vn])

    h = 1e-5
    J = np.stack([(f(h * e) - f(-h * e)) / (2 * h) for e in np.eye(12)], axis=1)
    assert np.max(np.abs(A - J)) < 5e-6, np.max(np.abs(A - J))


# ---- a9: the shim's control flow against the oracle's restatement of iLQR::Iteration ---------------------------------------
def _replay_trace(res, max_iter, min_iter):
    """Feeds the cost sequences the shim saw to the oracle's a9 functions (orc_update_lambda :636-657, orc_linesearch_accept
    :490-528, orc_check_convergence Optimiser.cpp:30-37) and demands the same lambda trajectory, acceptance decisions,
    best alpha and iteration count (:319-340)."""
    lam, old_cost = 0.1, res["cost_history"][0]                      # Optimiser.h:239
    expected_iterations = 0
    derivs_next = True
    for i, row in enumerate(res["trace"]):
        expected_iterations += 1
        assert bool(row["derivatives"]) == derivs_next                # skip the derivatives after a rejected step (:419)
        assert row["lambda_in"] == lam, (i, row["lambda_in"], lam)
        exited = False
        for attempt in range(int(row["backward_passes"])):
            last = attempt == int(row["backward_passes"]) - 1
            valid = last and not bool(row["lambda_exit"])
            lam, exited = orc.update_lambda(lam, valid)
            assert exited == (last and bool(row["lambda_exit"]))
        assert lam == row["lambda_after_backward"]
        if exited:
            break
        assert row["old_cost"] == old_cost
        costs = row["rollout_costs"]
        assert len(costs) == 6 and not np.any(np.isnan(costs))       # the reference rolls every alpha out
        best, new_cost, accepted, lam = orc.linesearch_accept(costs, old_cost, lam)
        assert best == int(row["best"]) and accepted == bool(row["accepted"]) and new_cost == row["new_cost"], (i, best, row)
        assert lam == row["lambda_out"]
        conv = orc.check_convergence(old_cost, new_cost)
        assert conv == bool(row["converged"])
        assert res["cost_history"][i + 1] == new_cost
        derivs_next = accepted
        if accepted:
            old_cost = new_cost
        if conv and i >= min_iter:
            break
    assert expected_iterations == res["iterations"] == len(res["trace"])


@pytest.mark.gpu
@pytest.mark.parametrize("model,T,opts", [("acrobot", 100, ""), ("acrobot", 100, "+unfused"), ("acrobot", 60, "+adaptive_jerk"),
                                          ("floating_body", 80, "")])
def test_iteration_control_flow_matches_the_oracle(model, T, opts):
    M = host.Model(model)
    rng = np.random.default_rng(5)
    u0 = None if model == "acrobot" else 0.2 * rng.standard_normal((T, M.nu))
    for max_iter, min_iter in ((8, 2), (4, 0)):
        res = host.optimise(model, T=T, max_iter=max_iter, min_iter=min_iter, options=opts, u_init=u0)
        assert len(res["trace"]) >= 1
        _replay_trace(res, max_iter, min_iter)
        h = res["cost_history"]
        assert np.all(np.diff(h) <= 1e-12)          # never worse; a run that only rejects (lambda exit) is a valid case


@pytest.mark.gpu
def test_free_joint_model_optimises_and_pruned_line_search_is_an_option():
    """Floating body (free joint: tangent-space FD columns and state feedback): the cost falls from the first iteration on; the
    GPU-ordered line search (opt-in) accepts only improving steps too but may stop at another alpha."""
    rng = np.random.default_rng(5)
    u0 = 0.2 * rng.standard_normal((80, 3))
    ref = host.optimise("floating_body", T=80, max_iter=10, min_iter=2, u_init=u0)
    assert ref["cost_history"][-1] < 0.9 * ref["cost_history"][0], ref["cost_history"]      # 0.8 s at 4 N: a fifth of the cost
    pr = host.optimise("floating_body", T=80, max_iter=10, min_iter=2, options="+pruned", u_init=u0)
    assert np.all(np.diff(pr["cost_history"]) <= 1e-12)
    for row in pr["trace"]:
        tried = row["rollout_costs"][~np.isnan(row["rollout_costs"])]
        assert len(tried) >= 1 and (not row["accepted"] or row["new_cost"] == tried.min() or row["new_cost"] in tried)


# ---- f3: iLQR_SVR's DoF importance (both branches) on the host ---------------------------------------------------------------
@pytest.mark.parametrize("dof,m,T,s", [(7, 7, 60, 1), (10, 7, 45, 4), (6, 3, 30, 2), (2, 1, 20, 1)])
def test_svr_dof_importance_host_both_branches(dof, m, T, s):
    rng = np.random.default_rng(dof * 100 + m)
    K = rng.standard_normal((T, 2 * dof, m)) * np.exp(rng.uniform(-3, 2, (1, 2 * dof, 1)))     # very different column scales
    sums, rem = host.dof_importance(K, dof, s, svd=False, threshold=0.0)
    assert np.array_equal(sums, orc.dof_importance(dof, m, T, s, K))
    ref = orc.dof_importance_svd(dof, m, T, s, K)
    got, _ = host.dof_importance(K, dof, s, svd=True)
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref)), np.max(np.abs(got - ref))
    thr = float(np.median(ref))
    _, rem = host.dof_importance(K, dof, s, svd=True, threshold=thr)
    assert list(rem) == [i for i in range(dof) if got[i] < thr]


# ---- the reference's own TestTasks rows as data fixtures for LoadTaskFromFile (FileHandler.cpp:471-578) --------------------
@pytest.mark.parametrize("task,n_start", [("acrobot", 2), ("piston_block", 1), ("push_ncl", 13), ("walker_run", 9)])
def test_load_task_from_the_references_test_task_files(task, n_start, golden_dir):
    """tests/golden/TestTasks/<task>/{0,7}.csv are rows copied from the reference's TestTasks/ (data, not code): start state
    (robot joints, then 6 numbers per rigid body) followed by the residual targets, a comma after every value."""
    import os
    for num in (0, 7):
        path = os.path.join(golden_dir, "TestTasks", task, f"{num}.csv")
        toks = [float(x) for x in open(path).read().strip().split(",") if x.strip()]
        n_targets = len(toks) - n_start
        assert n_targets > 0
        got = host.load_task(path, n_start, n_targets)
        assert got is not None
        assert np.array_equal(got[0], toks[:n_start]) and np.array_equal(got[1], toks[n_start:])
        assert host.load_task(path, n_start + 1, n_targets) is None        # wrong element count: refused (:519-523)


@pytest.mark.parametrize("model,stagger", [("acrobot", 0), ("acrobot", 1), ("floating_body", 0), ("floating_body", 2)])
def test_keypoint_ordered_fill_matches_the_job_lists(model, stagger):
    """The FD workers writing straight into the entry records of kpilqr_upload_fd_kp (Differentiator::DynamicsDerivativesKp)
    produce exactly the jobs of the job-list fill, slot by slot: x+ / x- rows, the nominal next state in the unstepped side of
    a one-sided control column (controls at their limits), the mode bits; uniform and ragged per-DoF key-point lists; hinge
    and free-joint (tangent-space position rows) models.  No GPU."""
    mdl = host.Model(model)
    T, nu = 60, mdl.nu
    rng = np.random.default_rng(7)
    lim = np.asarray(mdl.limits).reshape(nu, 2)
    u = rng.uniform(lim[:, 0], lim[:, 1], (T, nu)) * 0.2
    u[10:20] = lim[:, 1]                       # at the upper limit: backward-only control columns
    u[30:35] = lim[:, 0]                       # at the lower limit: forward-only
    r = host.fd_kp_check(model, T, 5, stagger, u)
    assert r["mismatches"] == 0, r
    assert r["one_sided"] > 0 and r["jobs"] > 0 and r["entries"] > 0

