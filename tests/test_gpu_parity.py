"""GPU parity tests: the HIP engine (through the C ABI, via trajoptkp_amd.Engine) against the CPU
oracle on identical seeded inputs.

Bars (BASELINE.json north_star / task brief):
  * fd_difference, interpolate, cost_derivs, and the generic (reference-order) backward / forward
    kernels: BIT-EXACT against the oracle (they are compiled with -ffp-contract=off and written in
    the reference's operation order; src/tests/Keypoints_Test.cpp:273-289 pins a4 bitwise).
  * MFMA backward pass: feedback gains K within 1e-6 relative (north_star); we assert 1e-9.
  * MFMA forward pass: predicted costs within 1e-9 relative, controls within 1e-9.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth

pytestmark = pytest.mark.gpu

K_RTOL = 1e-6          # north_star tolerance for the gains
K_RTOL_TIGHT = 1e-9    # what we actually hold the MFMA kernel to


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def run_engine(p, generic=False, pd_stride=100, lam=None, want_U=True):
    lam = p["lam"] if lam is None else lam
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], generic=generic) as e:
        synth.upload(e, p)
        e.fd_difference()
        e.sync()
        A_kp, B_kp = e.get_AB()
        e.interpolate()
        A, B = e.get_AB()
        e.cost_derivs()
        l_x, l_xx, l_u, l_uu = e.get_cost_derivs()
        status, dJ = e.backward(lam, pd_stride)
        K, k = e.gains()
        cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        return dict(A_kp=A_kp, B_kp=B_kp, A=A, B=B, l_x=l_x, l_xx=l_xx, l_u=l_u, l_uu=l_uu, status=status,
                    delta_J=dJ, K=K, k=k, cost_pred=cost, U_alpha=U,
                    variants=(e.backward_variant, e.forward_variant))


PROBLEMS = {
    "panda_T64": dict(task="panda_reaching", T=64, batch=2, min_N=5, config_id=2, dense_residuals=True, one_sided_frac=0.15),
    "acrobot_T100": dict(task="acrobot", T=100, batch=1, min_N=5, config_id=1, dense_residuals=True),
    "pushing_T48": dict(task="panda_pushing", T=48, batch=1, min_N=4, config_id=3, dense_residuals=True),
    "panda_T300_b3": dict(task="panda_reaching", T=300, batch=3, min_N=5, config_id=2),
}


@pytest.fixture(scope="module", params=list(PROBLEMS))
def case(request):
    p = synth.make_problem(**PROBLEMS[request.param])
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(p["batch"])]
    return request.param, p, ref


def test_elementwise_stages_bit_exact(case):
    name, p, ref = case
    g = run_engine(p, generic=True)
    for b, o in enumerate(ref):
        for key in ("A", "B", "l_x", "l_xx", "l_u", "l_uu"):
            assert np.array_equal(g[key][b], o[key]), f"{name} b{b} {key} not bit-exact"
        # key-point columns right after fd_difference
        kp_t = p["kp_times"]
        assert np.array_equal(g["A_kp"][b][kp_t], o["A_kp"][kp_t])
        assert np.array_equal(g["B_kp"][b][kp_t], o["B_kp"][kp_t])


def test_generic_backward_forward_bit_exact(case):
    name, p, ref = case
    g = run_engine(p, generic=True)
    assert g["variants"] == ("generic_lds", "generic_lds")
    for b, o in enumerate(ref):
        assert g["status"][b] == o["status"] == 0
        assert np.array_equal(g["K"][b], o["K"]), f"{name} b{b}: generic K differs: {relerr(g['K'][b], o['K'])}"
        assert np.array_equal(g["k"][b], o["k"])
        assert g["delta_J"][b] == o["delta_J"]
        assert np.array_equal(g["cost_pred"][b], o["cost_pred"])
        assert np.array_equal(g["U_alpha"][b], o["U_alpha"])


def test_mfma_backward_gains_within_tolerance(case):
    name, p, ref = case
    g = run_engine(p, generic=False)
    if p["dof"] * 2 + 1 > 16:
        assert g["variants"][0] == "mfma_f64_tiled", g["variants"]
    else:
        assert g["variants"][0] == "mfma_f64_t1", g["variants"]
    for b, o in enumerate(ref):
        assert g["status"][b] == 0
        eK, ek = relerr(g["K"][b], o["K"]), relerr(g["k"][b], o["k"])
        assert eK < K_RTOL and ek < K_RTOL, (name, b, eK, ek)
        assert eK < K_RTOL_TIGHT and ek < K_RTOL_TIGHT, (name, b, eK, ek)
        assert abs(g["delta_J"][b] - o["delta_J"]) <= 1e-9 * abs(o["delta_J"]) + 1e-300


def test_mfma_forward_within_tolerance(case):
    name, p, ref = case
    g = run_engine(p, generic=False)
    for b, o in enumerate(ref):
        scale = np.max(np.abs(o["cost_pred"]))
        assert np.max(np.abs(g["cost_pred"][b] - o["cost_pred"])) <= 1e-9 * scale, (name, b, g["cost_pred"][b], o["cost_pred"])
        assert relerr(g["U_alpha"][b], o["U_alpha"]) < 1e-9


def test_golden_fixtures(golden_dir):
    from oracle.crosscheck import GOLDEN
    for name, kw in GOLDEN.items():
        gold = np.load(f"{golden_dir}/{name}.npz")
        p = synth.make_problem(**kw)
        g = run_engine(p, generic=False)
        ge = run_engine(p, generic=True)
        for b in range(p["batch"]):
            for key in ("A", "B", "l_x", "l_xx", "l_u", "l_uu"):
                assert np.array_equal(g[key][b], gold[f"b{b}_{key}"]), (name, key)
            assert np.array_equal(ge["K"][b], gold[f"b{b}_K"])
            assert relerr(g["K"][b], gold[f"b{b}_K"]) < K_RTOL_TIGHT
            assert relerr(g["k"][b], gold[f"b{b}_k"]) < K_RTOL_TIGHT
            assert relerr(g["cost_pred"][b], gold[f"b{b}_cost_pred"]) < 1e-9


def test_full_size_panda_T3000(golden_dir):
    """BASELINE configs[1]: Panda reaching, T=3000, set-interval 5, batch 1 -- against the committed
    checksums and the oracle run here."""
    from oracle.crosscheck import GOLDEN_BIG
    kw = GOLDEN_BIG["panda_T3000"]
    gold = np.load(f"{golden_dir}/panda_T3000.npz")
    p = synth.make_problem(**kw)
    o = pipeline.run_trajectory(p, 0)
    g = run_engine(p, generic=False)
    assert g["status"][0] == 0
    for key in ("A", "B", "l_xx"):
        assert np.sum(g[key][0]) == float(gold[f"sum_{key}"])
        assert np.sum(np.abs(g[key][0])) == float(gold[f"abssum_{key}"])
    assert relerr(g["K"][0], o["K"]) < K_RTOL_TIGHT
    assert relerr(g["K"][0][0], gold["K_first"]) < K_RTOL_TIGHT
    assert relerr(g["K"][0][1500], gold["K_mid"]) < K_RTOL_TIGHT
    assert relerr(g["k"][0], o["k"]) < K_RTOL_TIGHT
    assert abs(g["delta_J"][0] - float(gold["delta_J"])) < 1e-9 * abs(float(gold["delta_J"]))
    assert relerr(g["cost_pred"][0], gold["cost_pred"]) < 1e-9
    # the generic kernel at full size: bit-exact with the oracle
    ge = run_engine(p, generic=True)
    assert np.array_equal(ge["K"][0], o["K"])


def test_batch_independence_and_sharding():
    """Trajectories are independent units: a batch of replicas gives identical results per replica,
    and results do not depend on which other trajectories share the batch (multi-GPU sharding)."""
    p1 = synth.make_problem(task="panda_reaching", T=200, batch=2, min_N=5)
    g1 = run_engine(p1)
    pt = synth.tile_problem(p1, 5)
    gt = run_engine(pt)
    for rep in range(5):
        for b in range(2):
            assert np.array_equal(gt["K"][rep * 2 + b], g1["K"][b])
            assert np.array_equal(gt["cost_pred"][rep * 2 + b], g1["cost_pred"][b])
    # shard: trajectory 1 alone
    p_single = synth.make_problem(task="panda_reaching", T=200, batch=1, min_N=5, first_b=1)
    gs = run_engine(p_single)
    assert np.array_equal(gs["K"][0], g1["K"][1])


def test_interpolation_known_answer_relation():
    """Interpolate.basic_interpolation (src/tests/Keypoints_Test.cpp:204-308): set_interval min_N=3,
    T=100: A[1] == A[0] + (A[3]-A[0])/3 bitwise; A[98] ~= A[96] + 2 (A[99]-A[96])/3."""
    p = synth.make_problem(task="acrobot", T=100, batch=1, min_N=3, config_id=1)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=1) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate()
        A, B = e.get_AB()
    A, B = A[0], B[0]
    for M in (A, B):
        diff = (M[3] - M[0]) / 3.0
        assert np.array_equal(M[1], M[0] + diff)
        d2 = (M[99] - M[96]) / 3.0
        assert np.allclose(M[98], M[96] + d2 + d2, rtol=0, atol=1e-6)


def test_ragged_keypoints_per_dof():
    """adaptive-jerk style key-points: every DoF has its own key-point times."""
    T, dof = 120, 7
    p = synth.make_problem(task="panda_reaching", T=T, batch=2, min_N=1, dense_residuals=True)   # FD at every step
    rng = np.random.default_rng(7)
    rows = []
    for b in range(2):
        offs = np.zeros(T + 1, np.int32); cols = []
        for t in range(T):
            offs[t] = len(cols)
            if t == 0 or t == T - 1:
                cols.extend(range(dof))
            else:
                cols.extend([i for i in range(dof) if rng.uniform() < 0.25])
        offs[T] = len(cols)
        rows.append((offs, np.asarray(cols, np.int32)))
    p["kp_rows"] = rows
    g = run_engine(p, generic=False)
    for b in range(2):
        o = pipeline.run_trajectory(p, b, want_U=True)
        assert np.array_equal(g["A"][b], o["A"]) and np.array_equal(g["B"][b], o["B"])
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT


def test_pd_failure_status_and_lambda_retry():
    """Non-PD Q_uu + lambda I: status = t+1 at the first CHECKED step (every pd_stride-th), exactly
    where the reference's CheckMatrixPD (iLQR.cpp:587-595) would return false; raising lambda fixes it."""
    p = synth.make_problem(task="panda_reaching", T=64, batch=2, min_N=5, dense_residuals=True)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e, \
            Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, generic=True) as eg:
        for eng in (e, eg):
            synth.upload(eng, p)
            eng.fd_difference(); eng.interpolate(); eng.cost_derivs()
        l_x, l_xx, l_u, l_uu = e.get_cost_derivs()
        l_uu[1] -= 5.0 * np.eye(p["m"])[None]          # make trajectory 1 indefinite in u
        for eng in (e, eg):
            eng.set_cost_derivs(l_uu=l_uu)
        A, B = e.get_AB()
        for stride in (1, 10):
            st_o = [orc.backward(p["n"], p["m"], p["T"], A[b], B[b], l_x[b], l_xx[b], l_u[b], l_uu[b], 0.1, stride)[0]
                    for b in range(2)]
            assert st_o[0] == 0 and st_o[1] > 0
            for eng in (e, eg):
                st, _ = eng.backward(0.1, stride)
                assert list(st) == st_o, (stride, eng.backward_variant, st, st_o)
        # unchecked indefinite steps (stride larger than T): follows Eigen's pivoted LDLT
        o = orc.backward(p["n"], p["m"], p["T"], A[1], B[1], l_x[1], l_xx[1], l_u[1], l_uu[1], 0.1, 1000)
        st, _ = e.backward(0.1, 1000)
        K, k = e.gains()
        assert st[1] == 0 and o[0] == 0
        assert relerr(K[1], o[1]) < 1e-6
        # lambda retry (iLQR.cpp:435-442): a larger lambda makes it PD again
        st, _ = e.backward(10.0, 1)
        assert list(st) == [0, 0]


def test_per_trajectory_lambda():
    p = synth.make_problem(task="panda_reaching", T=80, batch=3, min_N=5)
    lams = np.array([0.1, 1.0, 1e-4])
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=3) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        e.backward(lams)
        K, _ = e.gains()
    for b in range(3):
        o = pipeline.run_trajectory(p, b, lam=float(lams[b]), stages=("fd", "interp", "cost", "bwd"))
        assert relerr(K[b], o["K"]) < K_RTOL_TIGHT


def test_clamp_active_in_forward():
    """Tight control limits so that the clamp of iLQR.cpp:883-889 is active."""
    p = synth.make_problem(task="panda_reaching", T=100, batch=1, min_N=5)
    p["ctrl_lim"] = np.stack([p["u_nom"][0].min(0) - 1e-3, p["u_nom"][0].max(0) + 1e-3], axis=1).reshape(-1)
    o = pipeline.run_trajectory(p, 0, want_U=True)
    g = run_engine(p)
    lim = p["ctrl_lim"]
    assert np.any(o["U_alpha"] == lim[1::2][None, None, :]) or np.any(o["U_alpha"] == lim[0::2][None, None, :])
    assert relerr(g["U_alpha"][0], o["U_alpha"]) < 1e-9
    assert relerr(g["cost_pred"][0], o["cost_pred"]) < 1e-9


def test_iterate_equals_staged_calls():
    p = synth.make_problem(task="panda_reaching", T=150, batch=4, min_N=5)
    g = run_engine(p)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=4) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        K, k = e.gains()
    assert np.array_equal(K, g["K"]) and np.array_equal(res["cost_pred"], g["cost_pred"])
    assert np.array_equal(res["delta_J"], g["delta_J"]) and np.all(res["status"] == 0)


def test_trajectory_cost_matches_cost_function():
    p = synth.make_problem(task="panda_reaching", T=90, batch=2, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
        synth.upload(e, p)
        c = e.trajectory_cost()
    for b in range(2):
        ref = 0.0
        for t in range(p["T"]):
            ref += orc.cost_function(p["r"][b, t], p["w_term"] if t == p["T"] - 1 else p["w_run"])
        assert c[b] == ref


def test_bad_arguments_fail_loudly():
    from trajoptkp_amd import KpilqrError
    with pytest.raises(KpilqrError):
        Engine(0, 7, 10, 14)
    with Engine(7, 7, 20, 14) as e:
        with pytest.raises(KpilqrError):
            e.interpolate()                                      # before set_keypoints
        # FD job indices are checked on the device: the bad job is skipped and the next sync reports it
        e.set_keypoints_rows([synth.keypoint_rows_set_interval(7, 20, 5)])
        e.upload_fd([0], [25], [0], [0], np.zeros((1, 14)), np.zeros((1, 14)))   # t out of range
        e.fd_difference()
        with pytest.raises(KpilqrError):
            e.sync()
        e.upload_fd([0], [5], [0], [1], np.zeros((1, 14)), np.zeros((1, 14)))    # one-sided, no nominal
        e.fd_difference()
        with pytest.raises(KpilqrError):
            e.sync()
        with pytest.raises(KpilqrError):
            e.upload_fd([0], [5], [0], [0], np.zeros((1, 14)), np.zeros((1, 14)), eps=0.0)


@pytest.mark.parametrize("T", [2, 3, 7])
def test_tiny_horizons(T):
    """Smallest horizons the reference's loops admit (T >= 2: rows 0 and T-1 are always key-points)."""
    p = synth.make_problem(task="panda_reaching", T=T, batch=2, min_N=5, dense_residuals=True)
    g = run_engine(p)
    ge = run_engine(p, generic=True)
    for b in range(2):
        o = pipeline.run_trajectory(p, b, want_U=True)
        assert np.array_equal(g["A"][b], o["A"]) and np.array_equal(g["l_xx"][b], o["l_xx"])
        assert np.array_equal(ge["K"][b], o["K"])
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT
        assert relerr(g["cost_pred"][b], o["cost_pred"]) < 1e-9


def test_three_alphas_and_U_alpha_output():
    """n_alpha other than the reference's 6 (iLQR_SVR uses a different alpha set, iLQR_SVR.cpp:469-471)."""
    p = synth.make_problem(task="panda_reaching", T=50, batch=2, min_N=5, dense_residuals=True)
    alphas = np.array([1.0, 0.5, 0.1])
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, n_alpha=3) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate(); e.cost_derivs(); e.backward(p["lam"])
        cost, U = e.forward_linear(alphas, want_U=True)
    for b in range(2):
        o = pipeline.run_trajectory(p, b, stages=("fd", "interp", "cost", "bwd"))
        c, Uo = orc.forward_linear(p["n"], p["m"], p["T"], alphas, o["A"], o["B"], o["K"], o["k"], o["l_x"], o["l_xx"],
                                   o["l_u"], o["l_uu"], p["u_nom"][b], p["ctrl_lim"], want_U=True)
        assert relerr(cost[b], c) < 1e-9 and relerr(U[b], Uo) < 1e-9


def test_empty_fd_upload_and_rerun_is_idempotent():
    """Re-running a stage on unchanged inputs gives identical bytes; an empty FD upload is legal and
    leaves the records alone."""
    p = synth.make_problem(task="panda_reaching", T=40, batch=1, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=1) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6)); K1, k1 = e.gains(); A1, B1 = e.get_AB()
        e.iterate(); K2, k2 = e.gains(); A2, B2 = e.get_AB()
        assert np.array_equal(K1, K2) and np.array_equal(A1, A2) and np.array_equal(B1, B2)
        e.upload_fd([], [], [], [], np.zeros((0, p["n"])), np.zeros((0, p["n"])))
        e.fd_difference(); e.sync()
        A3, B3 = e.get_AB()
        assert np.array_equal(A1, A3)


def test_set_AB_and_cost_derivs_hooks_roundtrip():
    rng = np.random.default_rng(5)
    with Engine(3, 2, 9, 4, batch=2) as e:      # odd sizes: n=6, m=2 (tiled kernels through the catch-all num_ctrl instantiation)
        A = rng.standard_normal((2, 9, 6, 6)); B = rng.standard_normal((2, 9, 2, 6))
        lx = rng.standard_normal((2, 9, 6)); lxx = rng.standard_normal((2, 9, 6, 6))
        lu = rng.standard_normal((2, 9, 2)); luu = rng.standard_normal((2, 9, 2, 2))
        e.set_AB(A, B); e.set_cost_derivs(lx, lxx, lu, luu)
        A2, B2 = e.get_AB(); g = e.get_cost_derivs()
        assert np.array_equal(A, A2) and np.array_equal(B, B2)
        for x, y in zip((lx, lxx, lu, luu), g):
            assert np.array_equal(x, y)
        assert e.backward_variant == "mfma_f64_tiled"


@pytest.mark.parametrize("task,T,batch", [("panda_pushing", 60, 2), ("high_dof_push", 24, 2), ("panda_pushing", 301, 3)])
def test_tiled_mfma_large_state(task, T, batch):
    """n = 20 (2x2 tiles) and n = 62 (4x4 tiles): tiled MFMA backward pass (NT waves per trajectory, column
    decomposition) and forward pass against the oracle, and the generic kernel bit-exact beside it."""
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=4, dense_residuals=True, one_sided_frac=0.1)
    g = run_engine(p)
    ge = run_engine(p, generic=True)
    assert g["variants"] == ("mfma_f64_tiled", "mfma_f64_tiled"), g["variants"]
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, want_U=True)
        assert relerr(g["U_alpha"][b], o["U_alpha"]) < 1e-9
        assert g["status"][b] == 0
        assert np.array_equal(ge["K"][b], o["K"])
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT, relerr(g["K"][b], o["K"])
        assert relerr(g["k"][b], o["k"]) < K_RTOL_TIGHT
        assert abs(g["delta_J"][b] - o["delta_J"]) <= 1e-9 * abs(o["delta_J"])
        assert relerr(g["cost_pred"][b], o["cost_pred"]) < 1e-9


@pytest.mark.parametrize("task,T,batch", [("walker", 150, 2), ("arm8", 120, 2), ("arm5x2", 100, 2)])
def test_tiled_mfma_other_control_dims(task, T, batch):
    """num_ctrl other than 1 and 7 (walker 6, an 8-joint arm, a 5-joint arm with 2 motors) run on the tiled MFMA
    kernels through the padded catch-all instantiation of the backward kernel, small states with two tiles."""
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=3, dense_residuals=True, one_sided_frac=0.1)
    g = run_engine(p)
    # (the one-tile forward kernel takes any n + 2 <= 16, m <= 8; larger states use the tiled forward kernel)
    assert g["variants"][0] == "mfma_f64_tiled" and g["variants"][1] in ("mfma_f64_tiled", "mfma_f64_t1"), g["variants"]
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, want_U=True)
        assert g["status"][b] == 0
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT, relerr(g["K"][b], o["K"])
        assert relerr(g["k"][b], o["k"]) < K_RTOL_TIGHT
        assert abs(g["delta_J"][b] - o["delta_J"]) <= 1e-9 * abs(o["delta_J"])
        assert relerr(g["cost_pred"][b], o["cost_pred"]) < 1e-9
        assert relerr(g["U_alpha"][b], o["U_alpha"]) < 1e-9
    # indefinite Quu + lambda I on a checked step is reported at the same step as by the oracle
    p["w_run"] = -np.abs(p["w_run"]) - 1.0
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=batch) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, _ = e.backward(1e-4, 1)
    o = [pipeline.run_trajectory(p, b, lam=1e-4, pd_stride=1, stages=("fd", "interp", "cost", "bwd"))["status"] for b in range(batch)]
    assert list(st) == o and all(v > 0 for v in o)


@pytest.mark.parametrize("task,T,batch", [("hopper", 150, 2), ("pentabot", 100, 3)])
def test_one_tile_shapes_of_the_other_task_plugins(task, T, batch, wave_form):
    """hopper / floating cube (n=12, m=3) and pentabot (n=10, m=3) have one-tile instantiations like Panda and
    Acrobot: materialising kernels and fused sweeps (both wave organisations) against the oracle."""
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=3, dense_residuals=True, one_sided_frac=0.1)
    g = run_engine(p)
    assert g["variants"] == ("mfma_f64_t1", "mfma_f64_t1"), g["variants"]
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(batch)]
    check_fused(g, p, ref)
    check_fused(run_fused(p), p, ref)
    check_fused(run_fused(p, use_iterate=True), p, ref)


@pytest.mark.parametrize("task,T,batch", [("panda_pushing", 60, 2), ("high_dof_push", 24, 2), ("panda_pushing", 301, 3),
                                          ("walker", 150, 2), ("arm8", 120, 2), ("light_clutter_push", 77, 2)])
def test_tiled_a6_inside_the_sweeps(task, T, batch, monkeypatch):
    """KPILQR_FLAG_FUSED on a tiled shape: a6 (cost derivatives formed inside the tiled sweeps from the residuals and their
    Jacobians: kpilqr_cost_derivs is not run) -- against the oracle, with dense residual Jacobians incl. r_u, one-sided FD
    columns, terminal weights, and the PD-failure step.  (KPILQR_TILED_A6 forces the form for the small cases here; by itself
    the library takes it only for four-tile states at ~100 trajectories up.  The a4 form of rounds 2-4 was removed in round 5.)"""
    a6 = "1"
    monkeypatch.setenv("KPILQR_TILED_A6", a6)
    want = "mfma_f64_tiled_a6"
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=4, dense_residuals=True, one_sided_frac=0.1)
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(batch)]
    for use_iterate in (False, True):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=batch, fused=True) as e:
            assert (e.backward_variant, e.forward_variant) == (want, want)
            synth.upload(e, p)
            if use_iterate:
                e.iterate(p["lam"], 100, orc.alphas(6))
                res = e.results()
                g = dict(status=res["status"], delta_J=res["delta_J"], cost_pred=res["cost_pred"], U_alpha=None)
                g["K"], g["k"] = e.gains()
            else:
                e.fd_difference()
                e.interpolate()
                st, dJ = e.backward(p["lam"], 100)
                K, k = e.gains()
                cost, U = e.forward_linear(orc.alphas(6), want_U=True)
                g = dict(status=st, delta_J=dJ, K=K, k=k, cost_pred=cost, U_alpha=U)
        check_fused(g, p, ref)
    q = dict(p); q["w_run"] = -np.abs(p["w_run"]) - 1.0
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=batch, fused=True) as e:
        synth.upload(e, q)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, _ = e.backward(1e-4, 1)
    o = [pipeline.run_trajectory(q, b, lam=1e-4, pd_stride=1, stages=("fd", "interp", "cost", "bwd"))["status"] for b in range(batch)]
    assert list(st) == o and all(v > 0 for v in o)


# ---- fused sweeps (KPILQR_FLAG_FUSED): a4 + a6 evaluated inside the backward / forward kernels --------------
@pytest.fixture(params=["auto", "one_wave"])
def wave_form(request, monkeypatch):
    """The fused sweeps choose their wave organisation from the batch size (wave pairs per trajectory up to #SIMDs/2
    trajectories, one wave per trajectory beyond): the small parity cases would only ever see the pairs, so they are
    also run with the one-wave kernels of the headline batch forced."""
    if request.param == "one_wave":
        monkeypatch.setenv("KPILQR_FUSED_WAVES", "1")
        monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    return request.param


def run_fused(p, pd_stride=100, lam=None, n_alpha=6, use_iterate=False):
    lam = p["lam"] if lam is None else lam
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], n_alpha=n_alpha, fused=True) as e:
        assert (e.backward_variant, e.forward_variant) == ("mfma_f64_t1_fused", "mfma_f64_t1_fused")
        synth.upload(e, p)
        if use_iterate:
            e.iterate(lam, pd_stride, orc.alphas(n_alpha))
            res = e.results()
            status, dJ, cost = res["status"], res["delta_J"], res["cost_pred"]
            K, k = e.gains()
            U = None
        else:
            e.fd_difference()
            status, dJ = e.backward(lam, pd_stride)
            K, k = e.gains()
            cost, U = e.forward_linear(orc.alphas(n_alpha), want_U=True)
        return dict(status=status, delta_J=dJ, K=K, k=k, cost_pred=cost, U_alpha=U)


def check_fused(g, p, ref=None):
    for b in range(p["batch"]):
        o = ref[b] if ref is not None else pipeline.run_trajectory(p, b, want_U=True)
        assert g["status"][b] == 0
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT, relerr(g["K"][b], o["K"])
        assert relerr(g["k"][b], o["k"]) < K_RTOL_TIGHT, relerr(g["k"][b], o["k"])
        assert abs(g["delta_J"][b] - o["delta_J"]) <= 1e-9 * abs(o["delta_J"]) + 1e-300
        scale = np.max(np.abs(o["cost_pred"]))
        assert np.max(np.abs(g["cost_pred"][b] - o["cost_pred"])) <= 1e-9 * scale, (g["cost_pred"][b], o["cost_pred"])
        if g["U_alpha"] is not None:
            assert relerr(g["U_alpha"][b], o["U_alpha"]) < 1e-9


def test_fused_sweeps_match_oracle(case, wave_form):
    name, p, ref = case
    if p["dof"] * 2 + 2 > 16:
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=True) as e:
            assert "fused" not in e.backward_variant          # unsupported shape: the flag is ignored
        return
    check_fused(run_fused(p), p, ref)
    check_fused(run_fused(p, use_iterate=True), p, ref)


def test_fused_ragged_keypoints_and_dense_keypoints(wave_form):
    """Per-DoF key-point lists (every lane walks its own list), and key-points at EVERY step."""
    T, dof = 120, 7
    p = synth.make_problem(task="panda_reaching", T=T, batch=2, min_N=1, dense_residuals=True)
    check_fused(run_fused(p), p)                                 # dense: a segment per step
    rng = np.random.default_rng(11)
    rows = []
    for b in range(2):
        offs = np.zeros(T + 1, np.int32); cols = []
        for t in range(T):
            offs[t] = len(cols)
            if t == 0 or t == T - 1:
                cols.extend(range(dof))
            else:
                cols.extend([i for i in range(dof) if rng.uniform() < (0.6 if b == 0 else 0.08)])
        offs[T] = len(cols)
        rows.append((offs, np.asarray(cols, np.int32)))
    p["kp_rows"] = rows
    check_fused(run_fused(p), p)


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 6, 7, 9, 13])
def test_fused_tiny_horizons(T, wave_form):
    """Horizons around the depth of the forward sweep's request pipeline (four register sets: every remainder of T mod 4,
    horizons shorter than the pipeline), with and without control residuals, key-points every 5 and every 2 steps."""
    if T == 1:
        pytest.skip("the reference's key-point generators need T >= 2")
    for dense, mn in ((True, 5), (False, 2)):
        p = synth.make_problem(task="panda_reaching", T=T, batch=2, min_N=mn, dense_residuals=dense)
        check_fused(run_fused(p), p)


def test_fused_full_size_panda_T3000(wave_form):
    p = synth.make_problem(task="panda_reaching", T=3000, batch=2, min_N=5)
    g = run_fused(p)
    gu = run_engine(p)
    for b in range(2):
        # against the unfused MFMA path on the device (same inputs, A/B/l_* materialised) ...
        assert relerr(g["K"][b], gu["K"][b]) < K_RTOL_TIGHT
        assert relerr(g["cost_pred"][b], gu["cost_pred"][b]) < 1e-9
    check_fused(g, p)                                            # ... and against the oracle


def test_fused_pd_failure_and_noncanonical_keypoints(wave_form):
    p = synth.make_problem(task="panda_reaching", T=64, batch=2, min_N=5, dense_residuals=True)
    p["w_run"] = p["w_run"].copy(); p["w_term"] = p["w_term"].copy()
    # negative control-residual weights make l_uu (hence Q_uu + lambda I) indefinite
    if np.any(p["r_u"] != 0):
        p["w_run"][:] = -np.abs(p["w_run"]) - 1.0
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
            synth.upload(e, p)
            e.fd_difference()
            st, _ = e.backward(1e-4, 1)
        o = [pipeline.run_trajectory(p, b, lam=1e-4, pd_stride=1, stages=("fd", "interp", "cost", "bwd"))["status"] for b in range(2)]
        assert list(st) == o and all(s > 0 for s in o)
    # key-point lists that do not start at 0 / end at T-1 are refused loudly by the fused sweeps
    from trajoptkp_amd.engine import KpilqrError
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        offs = np.arange(0, 2 * p["dof"] + 1, dtype=np.int32) * 2
        times = np.tile(np.array([1, p["T"] - 1], np.int32), 2 * p["dof"])
        e.set_keypoints(offs, times)
        with pytest.raises(KpilqrError):
            e.backward(0.1)


# ---- key-point placement on the device -----------------------------------------------------
def _kp_states(rng, dof, T):
    """Smooth-ish joint trajectories with a few velocity kinks (what makes the adaptive methods fire)."""
    t = np.arange(T)[:, None] * 0.01
    q = np.cumsum(rng.standard_normal((T, dof)) * 0.02, axis=0) + np.sin(t * rng.uniform(1, 9, dof))
    v = np.gradient(q, 0.01, axis=0)
    for _ in range(3):
        v[int(rng.integers(2, T - 2)):, int(rng.integers(0, dof))] += rng.uniform(-2, 2)
    return np.concatenate([q, v], axis=1)


@pytest.mark.parametrize("method", ["set_interval", "adaptive_jerk", "adaptive_accel", "velocity_change"])
@pytest.mark.parametrize("shape", [(7, 300, 3), (2, 100, 2), (10, 65, 2), (31, 129, 1), (7, 2, 2), (7, 3000, 2)])
def test_device_keypoint_generation_matches_oracle(method, shape):
    from trajoptkp_amd.engine import rows_to_dof_csr
    dof, T, B = shape
    rng = np.random.default_rng(dof * 1000 + T)
    X = np.stack([_kp_states(rng, dof, T) if T > 8 else rng.standard_normal((T, 2 * dof)) for _ in range(B)])
    min_N, max_N = int(rng.integers(1, 5)), int(rng.integers(5, 40))
    thr = rng.uniform(50.0, 4000.0, dof) if method == "adaptive_jerk" else rng.uniform(0.005, 0.2, dof) if method == "adaptive_accel" \
        else rng.uniform(0.5, 20.0, dof)
    rows = []
    for b in range(B):
        if method == "set_interval":
            rows.append(orc.kp_set_interval(dof, T, min_N))
        elif method == "adaptive_jerk":
            rows.append(orc.kp_adaptive_jerk(dof, T, min_N, max_N, thr, 0.01, X[b]))
        elif method == "adaptive_accel":
            rows.append(orc.kp_adaptive_accel(dof, T, min_N, max_N, thr, X[b]))
        else:
            rows.append(orc.kp_velocity_change(dof, T, min_N, max_N, thr, X[b]))
    o_ref, t_ref = rows_to_dof_csr(rows, dof, T)
    with Engine(dof, min(dof, 7), T, 3, batch=B, generic=(2 * dof + 2 > 64)) as e:
        e.upload_states(X)
        e.generate_keypoints(method, min_N, max_N, None if method == "set_interval" else thr, 0.01)
        o_dev, t_dev = e.get_keypoints()
    assert np.array_equal(o_dev, o_ref), (method, shape)
    assert np.array_equal(t_dev, t_ref), (method, shape)


def test_device_keypoints_feed_the_pipeline():
    """Lists generated on the device drive interpolate and the fused sweeps exactly like uploaded ones."""
    p = synth.make_problem(task="panda_reaching", T=120, batch=2, min_N=1, dense_residuals=True)     # FD columns everywhere
    rng = np.random.default_rng(5)
    X = np.stack([_kp_states(rng, p["dof"], p["T"]) for _ in range(2)])
    thr = np.full(p["dof"], 2.0)
    rows = [orc.kp_velocity_change(p["dof"], p["T"], 2, 15, thr, X[b]) for b in range(2)]
    p["kp_rows"] = rows
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(2)]
    for fused in (False, True):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=fused) as e:
            synth.upload(e, p, keypoints=False)
            e.upload_states(X)
            e.generate_keypoints("velocity_change", 2, 15, thr, 0.01)
            e.iterate(p["lam"], 100, orc.alphas(6))
            K, k = e.gains()
            res = e.results()
            if not fused:
                A, B = e.get_AB()
        for b in range(2):
            assert relerr(K[b], ref[b]["K"]) < K_RTOL_TIGHT
            assert np.max(np.abs(res["cost_pred"][b] - ref[b]["cost_pred"])) <= 1e-9 * np.max(np.abs(ref[b]["cost_pred"]))
            if not fused:
                assert np.array_equal(A[b], ref[b]["A"]) and np.array_equal(B[b], ref[b]["B"])


# ---- A-matrix filters, iLQR_SVR DoF importance and alpha set -----------------------------
@pytest.mark.parametrize("method,coefs", [("low_pass", [0.25]), ("FIR", [0.1, 0.15, 0.5, 0.15, 0.1]), ("FIR", [1.0])])
def test_filter_dynamics_bit_exact(method, coefs):
    p = synth.make_problem(task="panda_reaching", T=150, batch=2, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate()
        A0, B0 = e.get_AB()
        e.filter_dynamics(method, coefs)
        A1, B1 = e.get_AB()
    for b in range(2):
        assert np.array_equal(A1[b], orc.filter_dynamics(p["dof"], p["T"], method, coefs, A0[b])), method
        assert np.array_equal(B1[b], B0[b])
    from trajoptkp_amd.engine import KpilqrError
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        with pytest.raises(KpilqrError):
            e.filter_dynamics(method, coefs)


def test_svr_dof_importance_and_alphas():
    p = synth.make_problem(task="panda_reaching", T=200, batch=3, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=3, fused=True) as e:
        synth.upload(e, p)
        e.fd_difference()
        e.backward(p["lam"])
        K, k = e.gains()
        for s in (1, 5):
            imp = e.dof_importance(s)
            for b in range(3):
                assert np.array_equal(imp[b], orc.dof_importance(p["dof"], p["m"], p["T"], s, K[b])), s
        # iLQR_SVR's line-search set 1 - i/6 through the same forward kernel
        cost, U = e.forward_linear(orc.alphas_svr(6), want_U=True)
    for b in range(3):
        o = pipeline.run_trajectory(p, b, stages=("fd", "interp", "cost", "bwd"))
        c_ref, U_ref = orc.forward_linear(p["n"], p["m"], p["T"], orc.alphas_svr(6), o["A"], o["B"], o["K"], o["k"], o["l_x"], o["l_xx"],
                                          o["l_u"], o["l_uu"], p["u_nom"][b], p["ctrl_lim"], want_U=True)
        assert np.max(np.abs(cost[b] - c_ref)) <= 1e-9 * np.max(np.abs(c_ref))
        assert relerr(U[b], U_ref) < 1e-9


def test_fused_batch_independence_lambda_clamp_and_alphas(wave_form):
    """The scenario tests of the materialising pipeline, on the fused sweeps."""
    # replicas are bit-identical; a trajectory does not depend on its batch neighbours
    p1 = synth.make_problem(task="panda_reaching", T=200, batch=2, min_N=5)
    g1 = run_fused(p1)
    gt = run_fused(synth.tile_problem(p1, 5))
    for rep in range(5):
        for b in range(2):
            assert np.array_equal(gt["K"][rep * 2 + b], g1["K"][b])
            assert np.array_equal(gt["cost_pred"][rep * 2 + b], g1["cost_pred"][b])
    gs = run_fused(synth.make_problem(task="panda_reaching", T=200, batch=1, min_N=5, first_b=1))
    assert np.array_equal(gs["K"][0], g1["K"][1])
    # per-trajectory lambda, down to the reference's min_lambda
    p = synth.make_problem(task="panda_reaching", T=80, batch=3, min_N=5)
    lams = np.array([0.1, 10.0, 1e-4])
    g = run_fused(p, lam=lams)
    for b in range(3):
        o = pipeline.run_trajectory(p, b, lam=float(lams[b]), want_U=True)
        assert relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT and relerr(g["k"][b], o["k"]) < K_RTOL_TIGHT
        assert np.max(np.abs(g["cost_pred"][b] - o["cost_pred"])) <= 1e-9 * np.max(np.abs(o["cost_pred"]))
    # active clamp (iLQR.cpp:883-889)
    p = synth.make_problem(task="panda_reaching", T=100, batch=1, min_N=5)
    p["ctrl_lim"] = np.stack([p["u_nom"][0].min(0) - 1e-3, p["u_nom"][0].max(0) + 1e-3], axis=1).reshape(-1)
    o = pipeline.run_trajectory(p, 0, want_U=True)
    g = run_fused(p)
    assert np.any(o["U_alpha"] == p["ctrl_lim"][1::2][None, None, :]) or np.any(o["U_alpha"] == p["ctrl_lim"][0::2][None, None, :])
    assert relerr(g["U_alpha"][0], o["U_alpha"]) < 1e-9
    assert np.max(np.abs(g["cost_pred"][0] - o["cost_pred"])) <= 1e-9 * np.max(np.abs(o["cost_pred"]))
    # three alphas, acrobot shape
    p = synth.make_problem(task="acrobot", T=60, batch=2, min_N=3, config_id=1, dense_residuals=True)
    g = run_fused(p, n_alpha=3)
    for b in range(2):
        o = pipeline.run_trajectory(p, b, n_alpha=3, want_U=True)
        assert g["cost_pred"].shape == (2, 3)
        assert np.max(np.abs(g["cost_pred"][b] - o["cost_pred"])) <= 1e-9 * np.max(np.abs(o["cost_pred"]))
        assert relerr(g["U_alpha"][b], o["U_alpha"]) < 1e-9


def test_c_abi_linesearch_allreduce_single_rank():
    """kpilqr_allreduce_linesearch: local pack (valid trajectories only) and, with a 1-rank RCCL communicator, the
    all-reduce itself; both must equal the sums computed on the host and trajoptkp_amd.distributed's packing."""
    import torch
    from trajoptkp_amd import distributed as kd
    p = synth.make_problem(task="panda_reaching", T=64, batch=5, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=5) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        ref = np.concatenate([res["cost_pred"][res["status"] == 0].sum(0), [res["delta_J"][res["status"] == 0].sum(), float((res["status"] == 0).sum())]])
        v0 = e.allreduce_linesearch()                       # no communicator: this rank's sums
        assert np.allclose(v0, ref, rtol=1e-13) and v0[7] == 5.0
        e.comm_init(1, 0, e.comm_unique_id())               # RCCL, one rank
        v1 = e.allreduce_linesearch()
        assert np.array_equal(v1, v0)
        cost = torch.as_tensor(res["cost_pred"]); dJ = torch.as_tensor(res["delta_J"]); st = torch.as_tensor(res["status"])
        assert np.allclose(kd.pack_linesearch(cost, dJ, st).numpy(), v0, rtol=1e-13)


@pytest.mark.parametrize("waves,fwd", [("1", "1"), ("5", "2"), ("5", "3")])
def test_fused_two_wave_backward_variant(monkeypatch, waves, fwd):
    """The wave organisations of the fused backward pass compute the same gains:
    KPILQR_FUSED_WAVES=1 one wave per trajectory (the default above #SIMDs/2 trajectories), =5 the consumer / helper pair
    (the default up to #SIMDs/2 trajectories).  (The control / state split, the producer / consumer pair and the triple of
    rounds 2-4 were removed in round 5: slower than the helper pair at every batch size.)"""
    monkeypatch.setenv("KPILQR_FUSED_WAVES", waves)
    # forward sweep: one wave per trajectory with "1", the state / cost pair with "2", the state / cost / staging triple
    # (the default up to #SIMDs/4 trajectories) with "3"
    monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", fwd)
    if waves == "5":
        monkeypatch.setenv("KPILQR_ROLE_SHIFT", "0")             # alternate the roles with the block index
    for kw in (PROBLEMS["panda_T64"], PROBLEMS["acrobot_T100"], dict(task="panda_reaching", T=301, batch=3, min_N=4)):
        p = synth.make_problem(**kw)
        check_fused(run_fused(p), p)
    # PD failure is reported through the shared flag by both waves
    p = synth.make_problem(task="panda_reaching", T=64, batch=2, min_N=5, dense_residuals=True)
    p["w_run"] = -np.abs(p["w_run"]) - 1.0
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        synth.upload(e, p)
        e.fd_difference()
        st, _ = e.backward(1e-4, 1)
    o = [pipeline.run_trajectory(p, b, lam=1e-4, pd_stride=1, stages=("fd", "interp", "cost", "bwd"))["status"] for b in range(2)]
    assert list(st) == o and all(v > 0 for v in o)


def test_fused_long_horizon_ragged_keypoints(wave_form):
    """T=1500 with very different key-point densities per DoF (one DoF only at 0 and T-1: a single 1499-step
    segment; one DoF at every step), fused against the oracle and against the materialising kernels."""
    T, dof = 1500, 7
    p = synth.make_problem(task="panda_reaching", T=T, batch=2, min_N=1)
    rng = np.random.default_rng(3)
    dens = [0.0, 1.0, 0.5, 0.02, 0.2, 0.003, 0.9]
    rows = []
    for b in range(2):
        offs = np.zeros(T + 1, np.int32); cols = []
        for t in range(T):
            offs[t] = len(cols)
            if t == 0 or t == T - 1:
                cols.extend(range(dof))
            else:
                cols.extend([i for i in range(dof) if rng.uniform() < dens[(i + b) % dof]])
        offs[T] = len(cols)
        rows.append((offs, np.asarray(cols, np.int32)))
    p["kp_rows"] = rows
    g = run_fused(p)
    gu = run_engine(p)
    for b in range(2):
        assert relerr(g["K"][b], gu["K"][b]) < K_RTOL_TIGHT and relerr(g["cost_pred"][b], gu["cost_pred"][b]) < 1e-9
    check_fused(g, p)


@pytest.mark.parametrize("mix", ["uniform", "mixed", "uniform_different_lists"])
def test_fused_forward_uniform_keypoint_flag(monkeypatch, mix):
    """The one-wave fused forward sweep has a form for key-point sets in which every DoF of a trajectory shares one list
    (it interpolates the transposed operands directly) and a general form; a device flag picks between them.  All-uniform
    batches (also with a DIFFERENT list per trajectory), and a batch where one trajectory is ragged (general form for all)."""
    monkeypatch.setenv("KPILQR_FUSED_WAVES", "1")
    monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    T, dof, B = 257, 7, 3
    rng = np.random.default_rng(11)
    rows = []
    for b in range(B):
        step = 5 if mix == "uniform" else 3 + 2 * b
        offs = np.zeros(T + 1, np.int32); cols = []
        for t in range(T):
            offs[t] = len(cols)
            if t == 0 or t == T - 1 or t % step == 0:
                cols.extend(range(dof))
            elif mix == "mixed" and b == B - 1:
                cols.extend([i for i in range(dof) if rng.uniform() < 0.15 * (i + 1) / dof])
        offs[T] = len(cols)
        rows.append((offs, np.asarray(cols, np.int32)))
    p = synth.make_ragged_problem("panda_reaching", T, rows, config_id=2, dense_residuals=(mix == "mixed"))
    check_fused(run_fused(p), p)


@pytest.mark.parametrize("batch", [256, 512, 520])
def test_fused_wave_forms_at_their_batch_limits(batch):
    """The consumer / helper pair runs the backward sweep while 2 x batch <= #SIMDs (512), one wave per trajectory beyond; the
    forward triple while 4 x batch <= #SIMDs: the batches either side of the switches, a few trajectories of each against the
    oracle, replicas of a seed bit-identical."""
    p0 = synth.make_problem(task="panda_reaching", T=60, batch=8, min_N=4, dense_residuals=True)
    reps = (batch + 7) // 8
    p = synth.tile_problem(p0, reps)
    g = run_fused(p)
    ref = [pipeline.run_trajectory(p0, b, want_U=True) for b in range(8)]
    for b in (0, 7, 8 * (reps - 1) + 3):
        o = ref[b % 8]
        assert g["status"][b] == 0 and relerr(g["K"][b], o["K"]) < K_RTOL_TIGHT and relerr(g["cost_pred"][b], o["cost_pred"]) < 1e-9
    assert np.array_equal(g["K"][3], g["K"][8 * (reps - 1) + 3])


def test_fused_indefinite_quu_on_unchecked_steps(wave_form):
    """Q_uu + lambda I indefinite while no PD check is due (pd_stride > T): the reference inverts it anyway with
    Eigen's pivoted LDLT (iLQR.cpp:597-604).  The fused backward pass (running-inverse fast path, LDL' fallback,
    pivoted slow path) must land on the same gains."""
    p = synth.make_problem(task="panda_reaching", T=64, batch=2, min_N=5, dense_residuals=True)
    assert np.any(p["r_u"] != 0)
    p["w_run"] = p["w_run"].copy(); p["w_term"] = p["w_term"].copy()
    p["w_run"][:] = -np.abs(p["w_run"]) - 1.0            # negative control-residual weights: l_uu indefinite
    for lam in (1e-4, 0.3):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
            synth.upload(e, p)
            e.fd_difference()
            st, _ = e.backward(lam, 1000)
            K, k = e.gains()
        for b in range(2):
            o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=1000, stages=("fd", "interp", "cost", "bwd"))
            assert st[b] == 0 and o["status"] == 0
            assert relerr(K[b], o["K"]) < 1e-6 and relerr(k[b], o["k"]) < 1e-6, (lam, b, relerr(K[b], o["K"]))


@pytest.mark.parametrize("task,T", [("panda_pushing", 300), ("walker", 200), ("light_clutter_push", 150), ("high_dof_push", 100)])
@pytest.mark.parametrize("uw", ["1", "0"])
def test_tiled_running_inverse_on_smooth_residual_jacobians(task, T, uw, monkeypatch):
    """Residual Jacobians that move slowly along the trajectory (synth.smooth_residual_jacobians: the structure of the
    reference's tasks): the tiled backward kernels then run on the Newton-Schulz refresh of the running inverse on nearly every
    step -- with independently drawn Jacobians, as in the other tiled tests, every step factorises -- with the u-wave (two /
    three tiles), with the interleaved chains (four tiles) and on the round-2 kernels, materialised and with a6 inside."""
    monkeypatch.setenv("KPILQR_TILED_UW", uw)
    monkeypatch.setenv("KPILQR_TILED_A6", "1")
    p = synth.make_problem(task=task, T=T, batch=2, min_N=4, dense_residuals="smooth", one_sided_frac=0.1)
    assert not np.any(p["r_u"])
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(2)]
    g = run_engine(p)
    assert g["variants"][0] == "mfma_f64_tiled", g["variants"]
    check_fused(g, p, ref)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:      # a6 inside the sweeps
        assert e.backward_variant == "mfma_f64_tiled_a6"
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        g = dict(status=res["status"], delta_J=res["delta_J"], cost_pred=res["cost_pred"], U_alpha=None)
        g["K"], g["k"] = e.gains()
    check_fused(g, p, ref)


@pytest.mark.parametrize("task", ["clutter_n48", "clutter_n52", "clutter_n56", "high_dof_push"])
@pytest.mark.parametrize("residuals", ["smooth", True])
def test_four_tile_sweeps_at_every_chunk_count(task, residuals, monkeypatch):
    """The four-tile sweeps are instantiated per count of four-row chunks in the last row tile (1 .. 4: n = 48, 52, 56, 62) -- the
    forms whose requests go out one by one under the products (round 5).  Materialised and with a6 inside, residual Jacobians
    smooth (running inverse) and drawn per step (every step factorises), against the oracle."""
    monkeypatch.setenv("KPILQR_TILED_A6", "1")
    p = synth.make_problem(task=task, T=70, batch=2, min_N=4, dense_residuals=residuals, one_sided_frac=0.1)
    ref = [pipeline.run_trajectory(p, b, want_U=True) for b in range(2)]
    g = run_engine(p)
    assert g["variants"][0] == "mfma_f64_tiled", g["variants"]
    check_fused(g, p, ref)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:      # a6 inside the sweeps
        assert e.backward_variant == "mfma_f64_tiled_a6"
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        g = dict(status=res["status"], delta_J=res["delta_J"], cost_pred=res["cost_pred"], U_alpha=None)
        g["K"], g["k"] = e.gains()
    check_fused(g, p, ref)


@pytest.mark.parametrize("task,T", [("panda_pushing", 20), ("walker", 24), ("light_clutter_push", 20)])
@pytest.mark.parametrize("uw", ["1", "0"])
def test_tiled_indefinite_quu_on_unchecked_steps(task, T, uw, monkeypatch):
    """The same on the tiled backward kernels: with the u-wave (two and three tiles: the inverse of the pivoted slow path
    reaches the column waves as a tile) and without it (every wave solves for its own columns).  Short horizons: with an
    indefinite Q_uu the recursion amplifies rounding differences (at T=48 either kernel is 1e-2 from the oracle at lambda=1e-4)."""
    monkeypatch.setenv("KPILQR_TILED_UW", uw)
    p = synth.make_problem(task=task, T=T, batch=2, min_N=4, dense_residuals=True)
    assert np.any(p["r_u"] != 0)
    p["w_run"] = -np.abs(p["w_run"]) - 1.0               # negative control-residual weights: l_uu indefinite
    for lam in (1e-4, 0.3):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
            assert e.backward_variant == "mfma_f64_tiled"
            synth.upload(e, p)
            e.fd_difference(); e.interpolate(); e.cost_derivs()
            st, _ = e.backward(lam, 1000)
            K, k = e.gains()
        for b in range(2):
            o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=1000, stages=("fd", "interp", "cost", "bwd"))
            assert st[b] == 0 and o["status"] == 0
            assert relerr(K[b], o["K"]) < 1e-6 and relerr(k[b], o["k"]) < 1e-6, (lam, b, relerr(K[b], o["K"]))


# ---- asynchronous boundary: pinned slab, device-side validation, chunk pipeline --------------------------
def _staged(p, fused):
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=fused) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results(); K, k = e.gains()
    return K, k, res


@pytest.mark.parametrize("task,T,batch,fused", [("panda_reaching", 200, 13, True), ("panda_reaching", 200, 13, False),
                                                ("panda_pushing", 90, 5, False), ("acrobot", 100, 3, True)])
def test_fd_slab_and_streamed_iteration_are_bit_identical(task, T, batch, fused):
    """The one-slab upload and the chunk-pipelined iteration give the bytes of the array-by-array upload + kpilqr_iterate,
    whatever the number of chunks."""
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=5, dense_residuals=True, one_sided_frac=0.2)
    K0, k0, res0 = _staged(p, fused)
    with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=fused) as e:
        synth.upload(e, p)                                    # key-points, weights, limits, nominal controls, residuals
        e.upload_fd([], [], [], [], np.zeros((0, p["n"])), np.zeros((0, p["n"])))     # forget the FD payload
        s = e.fd_slab(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], p["job_nom"], p["xnom"])
        e.upload_fd_slab(s, p["eps"])
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results(); K, k = e.gains()
    assert np.array_equal(K, K0) and np.array_equal(k, k0), task
    assert np.array_equal(res["cost_pred"], res0["cost_pred"]) and np.array_equal(res["delta_J"], res0["delta_J"])
    for nchunks in (1, 3, 4, 7, batch):
        with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=fused) as e:
            e.set_keypoints_rows(p["kp_rows"])
            e.upload_residuals(None, None, None, p["w_run"], p["w_term"])
            e.upload_nominal(None, p["ctrl_lim"])
            e.forward_linear(orc.alphas(6), fetch=False)         # alphas resident (the sweep itself runs on garbage once)
            s = e.fd_slab(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], p["job_nom"], p["xnom"])
            pin = {}
            for name in ("r", "r_x", "r_u", "u_nom"):
                pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
            lam = e.pinned(batch); lam[:] = p["lam"]
            K = e.pinned(K0.shape); k = e.pinned(k0.shape)
            cp = e.pinned((batch, 6)); dJ = e.pinned(batch); st = e.pinned(batch, np.int32)
            for _ in range(2):                                    # twice back to back: nothing waits in between
                e.iterate_streamed(fd=s, eps=p["eps"], lam=lam, K=K, k=k, cost_pred=cp, delta_J=dJ, status=st, nchunks=nchunks, **pin)
            e.sync()
            assert np.array_equal(K, K0) and np.array_equal(k, k0), (task, nchunks)
            assert np.array_equal(cp, res0["cost_pred"]) and np.array_equal(dJ, res0["delta_J"]) and np.all(st == 0)
            # and the ordinary calls after a streamed iteration see its results (the chunk streams are joined)
            K2, k2 = e.gains()
            assert np.array_equal(K2, K0)


def test_fd_indices_are_checked_on_the_device():
    """A job with an out-of-range trajectory, time, column, mode or nominal row is skipped by the kernel (no
    out-of-bounds write) and reported by the next kpilqr_sync; the valid jobs still land."""
    from trajoptkp_amd.engine import KpilqrError
    p = synth.make_problem(task="panda_reaching", T=40, batch=2, min_N=5, one_sided_frac=0.3)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
        synth.upload(e, p)
        e.fd_difference(); e.sync()
        A0, B0 = e.get_AB()
    for field, bad in (("job_b", 2), ("job_b", -1), ("job_t", 40), ("job_col", 21), ("job_mode", 3), ("job_nom", 10**6)):
        q = dict(p); q[field] = p[field].copy()
        j = int(np.nonzero(p["job_mode"] != 0)[0][0]) if field == "job_nom" else 5
        q[field][j] = bad
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
            e.set_keypoints_rows(q["kp_rows"])
            e.upload_fd(q["job_b"], q["job_t"], q["job_col"], q["job_mode"], q["xplus"], q["xminus"], job_nom=q["job_nom"], xnom=q["xnom"], eps=q["eps"])
            e.fd_difference()
            with pytest.raises(KpilqrError) as ei:
                e.sync()
            assert ei.value.code == -1 and "FD job" in str(ei.value), field
            e.sync()                                              # the flag is cleared once reported
            A, B = e.get_AB()
        # every other key-point is untouched by the bad job
        t_bad, b_bad = int(p["job_t"][j]), int(p["job_b"][j])
        mask = np.ones(A0.shape[:2], bool); mask[b_bad, t_bad] = False
        if field in ("job_b", "job_t"):
            mask[b_bad, t_bad:min(t_bad + 1, 40)] = False
        assert np.array_equal(A[mask], A0[mask]) and np.array_equal(B[mask], B0[mask]), field


# ---- f3: iLQR_SVR's variable state vector: kpilqr_resize re-uses the context (src/Optimiser/iLQR_SVR.cpp:38-193) -------------
def test_resize_in_place_reuses_allocations_and_matches_fresh_contexts():
    """Shrink the state vector, grow it back, change the control count and the horizon: after every kpilqr_resize the
    context gives the bytes of a freshly created one, and as long as the new sizes fit the device buffers stay put."""
    from trajoptkp_amd import _lib
    import ctypes as C
    shapes = [("panda_reaching", 200), ("pentabot", 150), ("panda_reaching", 200), ("acrobot", 100), ("panda_pushing", 120), ("hopper", 260)]

    def run(e, p):
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        return e.gains() + (res["cost_pred"], res["delta_J"], res["status"])

    def ptr(e, which):
        q, sz = C.c_void_p(), C.c_size_t()
        e._ck(e._L.kpilqr_device_ptr(e._h, which, C.byref(q), C.byref(sz)))
        return q.value, sz.value

    for fused in (True, False):
        first = synth.make_problem(task=shapes[0][0], T=shapes[0][1], batch=2, min_N=5, dense_residuals=True)
        with Engine(first["dof"], first["m"], first["T"], first["nr"], batch=2, fused=fused) as e:
            base_ptrs = None
            for task, T in shapes:
                p = synth.make_problem(task=task, T=T, batch=2, min_N=5, dense_residuals=True)
                if p["nr"] != first["nr"]:
                    p["nr"] = first["nr"]                         # the residual list does not change in a resize: pad it
                    pad = lambda a, ax: np.concatenate([a, np.zeros(a.shape[:ax] + (first["nr"] - a.shape[ax],) + a.shape[ax + 1:])], axis=ax)
                    p["r"] = pad(p["r"], 2); p["r_x"] = pad(p["r_x"], 2); p["r_u"] = pad(p["r_u"], 2)
                    p["w_run"] = pad(p["w_run"], 0); p["w_term"] = pad(p["w_term"], 0)
                e.resize(p["dof"], p["m"], p["T"])
                got = run(e, p)
                with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=fused) as f:
                    want = run(f, p)
                    assert e.backward_variant == f.backward_variant and e.forward_variant == f.forward_variant
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), (task, fused)
                ptrs = [ptr(e, w)[0] for w in (_lib.BUF_STEP_RECORDS, _lib.BUF_K, _lib.BUF_R_X)]
                if task == "panda_reaching":
                    base_ptrs = base_ptrs or ptrs
                    assert ptrs == base_ptrs                      # back at the first shape: same allocations
                elif task in ("pentabot", "acrobot"):
                    assert ptrs == base_ptrs                      # smaller shapes live in the first shape's buffers
            with pytest.raises(Exception):
                e.resize(0, 1, 10)
            e.resize(first["dof"], first["m"], first["T"])      # still usable after a refused resize
            assert np.array_equal(run(e, first)[0], run(e, first)[0])


# ---- f2: the error test of iterative_error on the device (KeyPointGenerator.cpp:550-640) ------------------------------------
@pytest.mark.parametrize("task,T,min_N,thr", [("panda_reaching", 400, 1, 3e-10), ("panda_reaching", 257, 5, 1e-9), ("high_dof_push", 300, 2, 1e-10)])
def test_iterative_error_bisection_with_the_device_error_test(task, T, min_N, thr):
    """GenerateKeyPointsIteratively (:449-548) driven level by level for a whole batch, the arithmetic of every level
    (CheckDOFColumnError) done by kpilqr_keypoint_error_test on the columns in the step records: the key-point sets are
    the oracle's (which bisects depth-first on the same dense A sequence), decision for decision."""
    cfg = synth.TASKS[task]
    dof, m, B = cfg["dof"], cfg["m"], 2
    n = 2 * dof
    dyn = [synth.dynamics_dense_smooth(np.random.default_rng(40 + b), dof, m, cfg["dt"], T) for b in range(B)]
    A = np.stack([d[0] for d in dyn]); Bm = np.stack([d[1] for d in dyn])
    with Engine(dof, m, T, cfg["nr"], batch=B) as e:
        e.set_AB(A, Bm)                       # every column "differenced": the host FD of a level is not what is tested here
        computed = set()
        pending = [(b, i, 0, T - 1) for b in range(B) for i in range(dof)]
        levels = 0
        while pending:
            levels += 1
            iv = np.asarray(pending, np.int32)
            good = e.keypoint_error_test(iv, min_N, thr)
            nxt = []
            for (b, i, s, en), g in zip(pending, good):
                if en - s <= min_N:
                    assert g
                    continue
                mid = (s + en) // 2
                computed.update([(b, i, s), (b, i, mid), (b, i, en)])
                if not g:
                    nxt += [(b, i, s, mid), (b, i, mid, en)]
            pending = nxt
        assert levels > 3
        from trajoptkp_amd.engine import KpilqrError
        with pytest.raises(KpilqrError):
            e.keypoint_error_test([[0, dof, 0, 5]], min_N, thr)
    for b in range(B):
        offs, cols = orc.kp_iterative_error(dof, T, min_N, thr, A[b])
        want = {(b, int(i), t) for t in range(T) for i in cols[offs[t]:offs[t + 1]]}
        got = {x for x in computed if x[0] == b}
        # the reference's generator adds full rows 0 and T-1 whatever the bisection found
        got |= {(b, i, 0) for i in range(dof)} | {(b, i, T - 1) for i in range(dof)}
        assert got == want, (task, b, len(got), len(want))
        assert 2 * dof < len(want) < T * dof
