"""GPU parity of the wide-control tiled sweeps (tiled_wide.hip): 8 < num_ctrl <= 32 -- the humanoid of
TaskConfigs/locomotion/humanoid.yaml (21 actuators) with and without its free root, a 12-actuator quadruped -- against the
CPU oracle: K, k, delta_J, predicted costs, controls; the PD-failure step of a checked step; an indefinite Q_uu + lambda I
on unchecked steps (Eigen's pivoted LDLT restated)."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def run(p, lam=None, pd=100):
    lam = p["lam"] if lam is None else lam
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"]) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, dJ = e.backward(lam, pd)
        K, k = e.gains()
        cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        return dict(status=st, delta_J=dJ, K=K, k=k, cost=cost, U=U, variants=(e.backward_variant, e.forward_variant))


@pytest.mark.parametrize("task,T,batch,variants", [("humanoid", 90, 2, ("mfma_f64_wide", "mfma_f64_wide")),
                                                  ("humanoid_fixed", 120, 2, ("mfma_f64_wide", "mfma_f64_wide")),
                                                  ("quadruped", 150, 3, ("mfma_f64_wide", "mfma_f64_tiled")),
                                                  ("humanoid", 301, 1, ("mfma_f64_wide", "mfma_f64_wide"))])
def test_wide_control_sweeps_match_oracle(task, T, batch, variants):
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=4, dense_residuals=True, one_sided_frac=0.1)
    g = run(p)
    assert g["variants"] == variants
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, want_U=True)
        assert g["status"][b] == 0 == o["status"]
        assert relerr(g["K"][b], o["K"]) < 1e-9, relerr(g["K"][b], o["K"])
        assert relerr(g["k"][b], o["k"]) < 1e-9
        assert abs(g["delta_J"][b] - o["delta_J"]) <= 1e-9 * abs(o["delta_J"])
        assert relerr(g["cost"][b], o["cost_pred"]) < 1e-9
        assert relerr(g["U"][b], o["U_alpha"]) < 1e-9


@pytest.mark.parametrize("task", ["humanoid_fixed", "quadruped"])
def test_wide_control_pd_failure_and_indefinite_steps(task):
    """Negative control-residual weights make l_uu (and with it Q_uu + lambda I) indefinite, as in
    test_gpu_parity.py::test_fused_indefinite_quu_on_unchecked_steps.  With a PD check due (pd_stride = 20) the first checked
    step reports it -- status = t + 1, the oracle's step; with none due (pd_stride > T) the reference inverts the matrix anyway
    with Eigen's pivoted LDLT (iLQR.cpp:597-604) and the sweep (running inverse on tiles, cooperative LDL' re-seeds, the
    pivoted slow path) must land on the same gains."""
    p = synth.make_problem(task=task, T=48, batch=2, min_N=5, dense_residuals=True)
    assert np.any(p["r_u"] != 0)
    p["w_run"] = -np.abs(p["w_run"]) - 1.0
    # lambda = 0.01: indefinite but not near-singular (l_uu has eigenvalues around -0.03; at 1e-4 the systems are within 1e-4 of
    # singular and a 48-step sweep amplifies ANY rounding difference to 1e-2: only the bit-exact generic kernels match there)
    for lam in (0.01, 0.3):
        g = run(p, lam=lam, pd=20)
        for b in range(2):
            o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=20, stages=("fd", "interp", "cost", "bwd"))
            assert g["status"][b] == o["status"]
            assert o["status"] > 0 or lam == 0.3
        g = run(p, lam=lam, pd=1000)
        for b in range(2):
            o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=1000, stages=("fd", "interp", "cost", "bwd"))
            assert g["status"][b] == o["status"] == 0
            assert relerr(g["K"][b], o["K"]) < 1e-6 and relerr(g["k"][b], o["k"]) < 1e-6, (lam, b, relerr(g["K"][b], o["K"]))


def test_wide_control_lambda_range_long_horizon():
    p = synth.make_problem(task="humanoid_fixed", T=700, batch=2, min_N=5)
    for lam in (1e-4, 10.0):
        g = run(p, lam=lam)
        for b in range(2):
            o = pipeline.run_trajectory(p, b, lam=lam, stages=("fd", "interp", "cost", "bwd"))
            assert g["status"][b] == o["status"] == 0
            assert relerr(g["K"][b], o["K"]) < 1e-9 and relerr(g["k"][b], o["k"]) < 1e-9
