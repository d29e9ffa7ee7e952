"""CPU tests of the oracle (oracle/kpilqr_oracle.c): against the committed golden vectors, against the
known-answer relations the reference's own tests hold for this path, and unit tests of the restated
Eigen pieces.  No GPU needed."""
import numpy as np
import pytest

from oracle import crosscheck, oracle as orc, pipeline
from trajoptkp_amd import synth


# ---- golden fixtures (tests/golden/*.npz, written by `python -m oracle.crosscheck --write`) ---------
@pytest.mark.parametrize("name", list(crosscheck.GOLDEN))
def test_oracle_reproduces_golden(name, golden_dir):
    gold = np.load(f"{golden_dir}/{name}.npz")
    p = synth.make_problem(**crosscheck.GOLDEN[name])
    for b in range(p["batch"]):
        o = pipeline.run_trajectory(p, b)
        assert o["status"] == 0
        for key in ("A", "B", "l_x", "l_xx", "l_u", "l_uu", "K", "k", "cost_pred"):
            assert np.array_equal(o[key], gold[f"b{b}_{key}"]), (name, b, key)
        assert o["delta_J"] == float(gold[f"b{b}_delta_J"])


def test_oracle_full_size_checksums(golden_dir):
    gold = np.load(f"{golden_dir}/panda_T3000.npz")
    p = synth.make_problem(**crosscheck.GOLDEN_BIG["panda_T3000"])
    o = pipeline.run_trajectory(p, 0)
    for key in ("A", "B", "l_xx", "K", "k", "cost_pred"):
        assert np.sum(o[key]) == float(gold[f"sum_{key}"]) and np.sum(np.abs(o[key])) == float(gold[f"abssum_{key}"])
    assert np.array_equal(o["K"][0], gold["K_first"]) and np.array_equal(o["K"][1500], gold["K_mid"])


@pytest.mark.parametrize("name", ["panda_T64", "acrobot_T100"])
def test_oracle_agrees_with_independent_numpy(name):
    _, _, worst = crosscheck.compare(name, crosscheck.GOLDEN[name], verbose=False)
    for key in ("A", "B", "l_x", "l_xx", "l_u", "l_uu"):
        assert worst[key] == 0.0, (key, worst[key])
    for key in ("K", "k", "delta_J", "cost_pred", "U_alpha"):
        assert worst[key] < 1e-10, (key, worst[key])


# ---- reference known-answer relations ----------------------------------------------------------------
def _acrobot_AB(min_N, T=100):
    p = synth.make_problem(task="acrobot", T=T, batch=1, min_N=min_N, config_id=1)
    o = pipeline.run_trajectory(p, 0, stages=("fd", "interp"))
    return p, o["A"], o["B"]


def test_interpolation_basic_relation_bitwise():
    """Interpolate.basic_interpolation, src/tests/Keypoints_Test.cpp:204-308: set_interval min_N=3,
    T=100: A[1] == A[0] + (A[3]-A[0])/3 BITWISE (ASSERT_EQ :273-289); A[98] ~ A[96] + 2(A[99]-A[96])/3
    within 1e-6 (:292-307).  Same for B."""
    _, A, B = _acrobot_AB(3)
    for M in (A, B):
        diff = (M[3] - M[0]) / 3.0
        assert np.array_equal(M[1], M[0] + diff)
        d2 = (M[99] - M[96]) / 3.0
        assert np.max(np.abs(M[98] - (M[96] + d2 + d2))) <= 1e-6


def _assert_keypoints(rows, dof, T, max_N):
    """AssertKeypoints, src/tests/Keypoints_Test.cpp:10-33."""
    assert rows[0][:dof] == list(range(dof))
    assert rows[T - 1][:dof] == list(range(dof))
    last = [0] * dof
    for t in range(T):
        for i in rows[t]:
            assert t - last[i] <= max_N
            last[i] = t


@pytest.mark.parametrize("min_N", [2, 3])
def test_keypoints_set_interval(min_N):
    """keypoints.set_interval, src/tests/Keypoints_Test.cpp:53-115: rows 0, min_N and T-1 are full."""
    dof, T = 2, 100
    offs, cols = orc.kp_set_interval(dof, T, min_N)
    rows = orc.kp_rows(offs, cols)
    assert len(rows) == T
    assert rows[0] == [0, 1] and rows[min_N] == [0, 1] and rows[T - 1] == [0, 1]
    assert rows[1] == []
    # twin used to build synthetic inputs agrees with the oracle
    o2, c2 = synth.keypoint_rows_set_interval(dof, T, min_N)
    assert np.array_equal(offs, o2) and np.array_equal(cols, c2)


def _free_fall_states(T=100, dt=0.01):
    """A smooth 2-DoF trajectory standing in for the acrobot free fall from (0.5, 0.1, 0, 0)."""
    t = np.arange(T) * dt
    q = np.stack([0.5 * np.cos(3.0 * t), 0.1 + 0.4 * np.sin(5.0 * t)], 1)
    v = np.stack([-1.5 * np.sin(3.0 * t), 2.0 * np.cos(5.0 * t)], 1)
    return np.concatenate([q, v], 1)


def test_keypoints_adaptive_jerk_structure():
    """keypoints.adaptive_jerk, src/tests/Keypoints_Test.cpp:117-159 (min_N 1, max_N 5)."""
    X = _free_fall_states()
    offs, cols = orc.kp_adaptive_jerk(2, 100, 1, 5, np.array([0.5, 0.5]), 0.01, X)
    _assert_keypoints(orc.kp_rows(offs, cols), 2, 100, 5)


def test_keypoints_adaptive_accel_structure():
    """adaptive_accel (KeyPointGenerator.cpp:98-101, 772-795): the reference's tests hold no case for it; the same structural
    asserts as keypoints.adaptive_jerk (rows 0 / T-1 full, max gap <= max_N) plus the definition of the profile: a key-point
    inside the horizon sits where v[t+1] - v[t] exceeds the threshold (signed) or where max_N forces one."""
    X = _free_fall_states()
    thr = np.array([0.01, 0.01])
    offs, cols = orc.kp_adaptive_accel(2, 100, 1, 5, thr, X)
    rows = orc.kp_rows(offs, cols)
    _assert_keypoints(rows, 2, 100, 5)
    acc = X[1:, 2:] - X[:-1, 2:]
    last = [0, 0]
    for t in range(1, 99):
        for j in range(2):
            expect = (t - last[j] >= 1 and acc[t, j] > thr[j]) or (t - last[j] >= 5)
            assert (j in rows[t]) == expect, (t, j)
            if expect:
                last[j] = t


def test_keypoints_velocity_change_structure():
    """keypoints.velocity_change, src/tests/Keypoints_Test.cpp:161-202 (min_N 1, max_N 5)."""
    X = _free_fall_states()
    offs, cols = orc.kp_velocity_change(2, 100, 1, 5, np.array([0.5, 0.5]), X)
    rows = orc.kp_rows(offs, cols)
    # the reference appends every DoF to the last row (KeyPointGenerator.cpp:724-727)
    assert rows[99][-2:] == [0, 1]
    last = [0, 0]
    for t in range(100):
        for i in rows[t]:
            assert t - last[i] <= 5
            last[i] = t


def test_keypoints_iterative_error_bisection():
    """GenerateKeyPointsIteratively (KeyPointGenerator.cpp:449-640) on a dense A sequence: a linear-in-t
    sequence needs only start/mid/end; a kink forces refinement around it."""
    dof, T = 2, 65
    n = 4
    A = np.zeros((T, n, n))
    for t in range(T):
        A[t] = np.eye(n) + 0.01 * t
    offs, cols = orc.kp_iterative_error(dof, T, 1, 1e-9, A)
    rows = orc.kp_rows(offs, cols)
    assert rows[0] == [0, 1] and rows[32] == [0, 1] and rows[64] == [0, 1]
    assert sum(len(r) for r in rows) == 6
    A2 = A.copy()
    A2[40:, 0, 2] += 0.5                      # DoF 0's column 0, a velocity row: step change at t=40
    offs, cols = orc.kp_iterative_error(dof, T, 1, 1e-9, A2)
    rows = orc.kp_rows(offs, cols)
    assert 0 in rows[39] or 0 in rows[40]
    assert sum(1 for r in rows if 0 in r) > 3 and sum(1 for r in rows if 1 in r) == 3
    pct = orc.kp_percentages(dof, T, offs, cols)
    assert pct[1] == pytest.approx(3 / 65 * 100)


def test_interpolate_with_duplicate_last_row_entries():
    """velocity_change can list a DoF twice in the last row; the reference's loop then sees a zero-length
    interval and writes nothing (KeyPointGenerator.cpp:896-949)."""
    p = synth.make_problem(task="acrobot", T=20, batch=1, min_N=4, config_id=1)
    o = pipeline.run_trajectory(p, 0, stages=("fd", "interp"))
    offs, cols = p["kp_rows"][0]
    cols2 = np.concatenate([cols, [0, 1]]).astype(np.int32)
    offs2 = offs.copy(); offs2[-1] += 2
    p2 = dict(p); p2["kp_rows"] = [(offs2, cols2)]
    o2 = pipeline.run_trajectory(p2, 0, stages=("fd", "interp"))
    assert np.array_equal(o["A"], o2["A"]) and np.array_equal(o["B"], o2["B"])


# ---- restated Eigen pieces ---------------------------------------------------------------------------
def test_ldlt_inverse_and_llt():
    rng = np.random.default_rng(0)
    for m in (1, 3, 7, 11):
        G = rng.standard_normal((m, m))
        S = G @ G.T + 0.1 * np.eye(m)
        assert orc.llt_is_pd(S)
        inv = orc.ldlt_inverse(S)
        assert np.max(np.abs(inv @ S - np.eye(m))) < 1e-10
        # indefinite but non-singular: LDLT still inverts, LLT reports failure
        D = S.copy(); D[0, 0] -= 50.0
        assert not orc.llt_is_pd(D)
        assert np.max(np.abs(orc.ldlt_inverse(D) @ D - np.eye(m))) < 1e-8
    # only the lower triangle is read
    S2 = S.copy(); S2[0, 1] += 123.0
    assert np.array_equal(orc.ldlt_inverse(np.tril(S2) + np.tril(S2, -1).T), orc.ldlt_inverse(np.tril(S) + np.tril(S, -1).T))


def test_backward_pd_check_stride():
    """CheckMatrixPD is consulted only every pd_stride-th step (iLQR.cpp:565,587-595)."""
    p = synth.make_problem(task="panda_reaching", T=64, batch=1, min_N=5, dense_residuals=True)
    o = pipeline.run_trajectory(p, 0, stages=("fd", "interp", "cost"))
    l_uu = o["l_uu"] - 5.0 * np.eye(7)[None]
    args = (14, 7, 64, o["A"], o["B"], o["l_x"], o["l_xx"], o["l_u"], l_uu, 0.1)
    assert orc.backward(*args, 1)[0] == 64          # first step (t = T-1) fails
    assert orc.backward(*args, 10)[0] == 55         # 10th step counted from T-1
    assert orc.backward(*args, 1000)[0] == 0        # never checked
    assert orc.backward(14, 7, 64, o["A"], o["B"], o["l_x"], o["l_xx"], o["l_u"], l_uu, 10.0, 1)[0] == 0


def test_terminal_weights_rewrite_last_step():
    """Optimiser::ComputeCostDerivatives recomputes t = T-1 with terminal weights (Optimiser.cpp:208-211)."""
    p = synth.make_problem(task="panda_reaching", T=16, batch=1, min_N=5, dense_residuals=True)
    o = pipeline.run_trajectory(p, 0, stages=("cost",))
    t = 15
    w = p["w_term"]
    lx = sum(2 * w[i] * p["r"][0, t, i] * p["r_x"][0, t, i] for i in range(p["nr"]))
    assert np.allclose(o["l_x"][t], lx, rtol=1e-13)
    w = p["w_run"]
    lx = sum(2 * w[i] * p["r"][0, 3, i] * p["r_x"][0, 3, i] for i in range(p["nr"]))
    assert np.allclose(o["l_x"][3], lx, rtol=1e-13)


# ---- a9: scalar control flow -------------------------------------------------------------------------
def test_lambda_schedule_and_convergence():
    lam, ex = orc.update_lambda(0.1, True)            # valid pass: lambda / 10   (iLQR.cpp:642-644)
    assert lam == pytest.approx(0.01) and not ex
    lam, ex = orc.update_lambda(0.00005, True)        # clamp at min_lambda
    assert lam == 1e-4 and not ex
    lam, ex = orc.update_lambda(5.0, False)           # failed pass: x10, exit above max_lambda
    assert lam == 10.0 and ex
    assert orc.check_convergence(100.0, 99.0)         # (old-new)/new < 0.02   (Optimiser.cpp:30-37)
    assert not orc.check_convergence(100.0, 90.0)
    best, new_cost, acc, lam = orc.linesearch_accept(np.array([5.0, 3.0, 4.0]), 3.5, 0.01)
    assert best == 1 and acc and new_cost == 3.0 and lam == 0.01
    best, new_cost, acc, lam = orc.linesearch_accept(np.array([5.0, 4.0, 4.5]), 3.5, 0.5)
    assert not acc and new_cost == 3.5 and lam == 10.0     # x100, clamped (iLQR.cpp:525-527)
    assert np.allclose(orc.alphas(6), [(i / 6) ** 2 for i in range(1, 7)])


def test_filters_and_svr_pieces_against_numpy():
    """SURVEY 8f helpers of the oracle against independent numpy statements of the same reference lines."""
    rng = np.random.default_rng(3)
    dof, m, T = 3, 2, 40
    n = 2 * dof
    A = rng.standard_normal((T, n, n))
    a = 0.25
    lp = orc.filter_dynamics(dof, T, "low_pass", [a], A)
    fir_c = [0.1, 0.15, 0.5, 0.15, 0.1]
    fir = orc.filter_dynamics(dof, T, "FIR", fir_c, A)
    # memory layout is column-major per matrix: element (row i, col j) of A[t] is A[t, j, i]
    for i in range(n):
        for j in range(n):
            x = A[:, j, i]
            if i < dof:
                assert np.array_equal(lp[:, j, i], x) and np.array_equal(fir[:, j, i], x)       # position rows untouched
                continue
            y = np.zeros(T); yn1 = xn1 = x[0]
            for k in range(T):
                yn = ((1 - a) * yn1) + a * ((x[k] + xn1) / 2)
                xn1, yn1 = x[k], yn
                y[k] = yn
            assert np.array_equal(lp[:, j, i], y)
            f = np.zeros(T)
            for k in range(T):
                for c, co in enumerate(fir_c):
                    if k - c >= 0:
                        f[k] += x[k - c] * co
            assert np.array_equal(fir[:, j, i], f)
    K = rng.standard_normal((T, n, m))                        # column-major m x n per step
    s = orc.dof_importance(dof, m, T, 3, K)
    ref = np.zeros(dof)
    for t in range(0, T, 3):
        for i in range(dof):
            for j in range(m):
                ref[i] += abs(K[t, i, j]); ref[i] += abs(K[t, i + dof, j])
    assert np.array_equal(s, ref / T)
    assert np.array_equal(orc.alphas_svr(6), 1.0 - np.arange(6) / 6)
