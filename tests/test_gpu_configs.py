"""GPU parity at the FULL workloads of BASELINE.json's configs[2], [3] and [4] (the small-size cases live in
test_gpu_parity.py).  Everything goes through the C ABI (trajoptkp_amd.Engine) and is held to the CPU oracle on the
same seeded inputs:

  configs[2]  Franka Panda pushing (n = 20, two tiles), T = 3000, batch = 64, key-points placed by the DEVICE
              adaptive_jerk generator on a synthetic contact trajectory -- ragged per-DoF lists;
  configs[3]  Panda reaching, T = 3000, batch = 1024 on one GPU: the fused one-wave-per-trajectory kernels of the
              headline number -- 8 distinct seeds against the oracle, every replica bit-identical to its seed;
  configs[4]  high-DoF push (n = 62, four tiles), T = 5000, iterative_error key-points emulated by the reference's
              bisection on a dense synthetic A sequence -- ragged per-DoF lists on the tiled kernels;
  plus ragged per-DoF lists on the two- and three-tile kernels at moderate horizons.

Bars: K, k within 1e-9 relative of the oracle (north star: 1e-6), delta_J and predicted costs within 1e-9.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth
from trajoptkp_amd.engine import rows_to_dof_csr

pytestmark = pytest.mark.gpu

TIGHT = 1e-9


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def check_against_oracle(p, b, K, k, res, lam=None):
    o = pipeline.run_trajectory(p, b, lam=lam)
    assert o["status"] == 0 and res["status"][b] == 0, (b, o["status"], res["status"][b])
    eK, ek = relerr(K[b], o["K"]), relerr(k[b], o["k"])
    assert eK < TIGHT and ek < TIGHT, (b, eK, ek)
    assert abs(res["delta_J"][b] - o["delta_J"]) <= TIGHT * abs(o["delta_J"]), (b, res["delta_J"][b], o["delta_J"])
    scale = np.max(np.abs(o["cost_pred"]))
    assert np.max(np.abs(res["cost_pred"][b] - o["cost_pred"])) <= TIGHT * scale, (b, res["cost_pred"][b], o["cost_pred"])
    return eK


def run_iteration(p, fused, keypoints=True, states=None, gen=None, expect=None):
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=fused) as e:
        if expect is not None:
            assert e.backward_variant in expect, (e.backward_variant, expect)
        synth.upload(e, p, keypoints=keypoints)
        if gen is not None:
            e.upload_states(states)
            e.generate_keypoints(*gen)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        K, k = e.gains()
        return K, k, res, e.backward_variant


# ---- configs[2] ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pushing_B64():
    dof, T, B, dt = 10, 3000, 64, 0.008
    thr = np.array([10.0] * 7 + [1.0] * 3)           # jerk thresholds: joints / pushed body
    X = np.stack([synth.contact_trajectory(np.random.default_rng(synth.seed_for(3, b) + 17), dof, T, dt) for b in range(B)])
    rows = [orc.kp_adaptive_jerk(dof, T, 1, 100, thr, dt, X[b]) for b in range(B)]
    p = synth.make_ragged_problem("panda_pushing", T, rows, config_id=3, dense_residuals=True, one_sided_frac=0.05)
    return p, X, thr, dt


@pytest.mark.parametrize("fused", [False, True])
def test_config2_pushing_T3000_B64_device_adaptive_jerk(pushing_B64, fused):
    p, X, thr, dt = pushing_B64
    dof, T, B = p["dof"], p["T"], p["batch"]
    # the device generator must place exactly the oracle's key-points ...
    with Engine(dof, p["m"], T, p["nr"], batch=B, fused=fused) as e:
        e.upload_states(X)
        e.generate_keypoints("adaptive_jerk", 1, 100, thr, dt)
        o_dev, t_dev = e.get_keypoints()
    o_ref, t_ref = rows_to_dof_csr(p["kp_rows"], dof, T)
    assert np.array_equal(o_dev, o_ref) and np.array_equal(t_dev, t_ref)
    counts = np.diff(o_ref).reshape(B, dof)
    assert counts.min() >= 31 and counts.max() > 300        # max_N gaps only ... clusters of hundreds: ragged lists
    # ... and drive the whole iteration from them (FD jobs only at those key-points)
    K, k, res, variant = run_iteration(p, fused, keypoints=False, states=X, gen=("adaptive_jerk", 1, 100, thr, dt))
    assert variant.startswith("mfma_f64_tiled"), variant
    assert np.all(res["status"] == 0)
    for b in (0, 1, 17, 31, 40, 63):
        check_against_oracle(p, b, K, k, res)


# ---- configs[3] ---------------------------------------------------------------------------------------------------
def test_config3_panda_B1024_fused_one_wave_kernels():
    uniq, reps, T = 8, 128, 3000
    p0 = synth.make_problem(task="panda_reaching", T=T, batch=uniq, min_N=5)
    p = synth.tile_problem(p0, reps)
    assert p["batch"] == 1024
    with Engine(p["dof"], p["m"], T, p["nr"], batch=1024, fused=True) as e:
        assert e.backward_variant == "mfma_f64_t1_fused"
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results()
        K, k = e.gains()
    assert np.all(res["status"] == 0)
    worst = 0.0
    for b in range(uniq):
        worst = max(worst, check_against_oracle(p0, b, K, k, res))
    # every replica is bit-identical to its seed (same inputs, deterministic kernels whatever the SIMD they land on)
    Kr = K.reshape(reps, uniq, *K.shape[1:]); kr = k.reshape(reps, uniq, *k.shape[1:])
    assert np.array_equal(Kr, np.broadcast_to(Kr[0], Kr.shape))
    assert np.array_equal(kr, np.broadcast_to(kr[0], kr.shape))
    for key in ("delta_J", "cost_pred"):
        v = res[key].reshape(reps, uniq, -1)
        assert np.array_equal(v, np.broadcast_to(v[0], v.shape)), key
    print(f"config3: worst K rel err over {uniq} seeds = {worst:.2e}")


# ---- configs[4] ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def high_dof_T5000():
    dof, T, B = 31, 5000, 2
    rows, dyn = [], []
    for b in range(B):
        rng = np.random.default_rng(synth.seed_for(5, b) + 77)
        A, Bm = synth.dynamics_dense_smooth(rng, dof, 7, 0.008, T)
        rows.append(orc.kp_iterative_error(dof, T, 1, 1e-11, A))       # the reference's bisection on the dense sequence
        dyn.append((A, Bm))
    pct = [orc.kp_percentages(dof, T, *r) for r in rows]
    p = synth.make_ragged_problem("high_dof_push", T, rows, dyn=dyn, config_id=5, dense_residuals=True)
    return p, pct


@pytest.mark.parametrize("a6", ["0", "1"])
def test_config4_high_dof_T5000_iterative_error_keypoints(high_dof_T5000, a6, monkeypatch):
    p, pct = high_dof_T5000
    for q in pct:                                   # 20-45 % key-points on average, very different per DoF
        assert 15.0 < q.mean() < 50.0 and q.min() < 1.0 and q.max() > 60.0, (q.mean(), q.min(), q.max())
    monkeypatch.setenv("KPILQR_TILED_A6", a6)
    K, k, res, variant = run_iteration(p, fused=True)
    assert variant.startswith("mfma_f64_tiled"), variant
    assert ("a6" in variant) == (a6 == "1"), variant
    for b in range(p["batch"]):
        check_against_oracle(p, b, K, k, res)


# ---- ragged per-DoF lists on the two- and three-tile kernels ------------------------------------------------------------
@pytest.mark.parametrize("task,T,batch", [("panda_pushing", 700, 3), ("light_clutter_push", 500, 2)])
@pytest.mark.parametrize("fused", [False, True])
def test_ragged_lists_on_two_and_three_tiles(task, T, batch, fused):
    dof = synth.TASKS[task]["dof"]
    rng = np.random.default_rng(dof * 7 + T)
    dens = rng.uniform(0.05, 0.95, dof)
    dens[0], dens[1] = 0.0, 1.0                      # one DoF only at 0 and T-1, one at (nearly) every step
    rows = [synth.bisect_keypoints(rng, dof, T, 1, np.roll(dens, b)) for b in range(batch)]
    p = synth.make_ragged_problem(task, T, rows, config_id=4, dense_residuals=True, one_sided_frac=0.1)
    K, k, res, variant = run_iteration(p, fused)
    assert variant.startswith("mfma_f64_tiled"), variant
    for b in range(batch):
        check_against_oracle(p, b, K, k, res)


@pytest.mark.parametrize("task,T,batch", [("panda_pushing", 420, 4), ("walker", 333, 3), ("arm5x2", 200, 2), ("arm8", 150, 2)])
def test_two_tile_forward_state_cost_wave_groups(task, T, batch, monkeypatch):
    """The two-tile forward sweep on materialised tiles comes in two organisations: one wave per row tile (k_forward_tiled), and --
    while the 2 NT waves of a trajectory each find a SIMD, the library's default then -- state / cost wave groups
    (k_forward_tiled_sc: the recursion's waves do not score, two more waves score one step behind).  Both against the oracle
    (control law / clamp iLQR.cpp:876-890 exact, predicted costs 1e-9) and against each other, controls included; the library
    says which one ran (kpilqr_last_launch)."""
    dof = synth.TASKS[task]["dof"]
    rng = np.random.default_rng(dof + T)
    rows = [synth.bisect_keypoints(rng, dof, T, 1, rng.uniform(0.05, 0.9, dof)) for _ in range(batch)]
    p = synth.make_ragged_problem(task, T, rows, config_id=4, dense_residuals=True, one_sided_frac=0.1)
    p["u_nom"] *= 3.0                                            # some candidates hit the control limits
    out = {}
    for fsc in ("0", "1", None):
        if fsc is None:
            monkeypatch.delenv("KPILQR_TILED_FSC", raising=False)
        else:
            monkeypatch.setenv("KPILQR_TILED_FSC", fsc)
        # (arm5x2, n = 10, fits one tile: KPILQR_FLAG_TILED_KERNELS runs it on two -- the second tile all structural zeros)
        with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, tiled=task == "arm5x2") as e:
            synth.upload(e, p)
            e.fd_difference(); e.interpolate(); e.cost_derivs()
            st, dJ = e.backward(p["lam"], 100)
            cost, U = e.forward_linear(orc.alphas(6), want_U=True)
            out[fsc] = (cost, U, e.last_launch("forward"))
            assert np.all(st == 0)
    assert out["0"][2] == "mfma_f64_tiled" and out["1"][2] == "mfma_f64_tiled:state_cost_waves" and out[None][2] == out["1"][2], [v[2] for v in out.values()]
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, want_U=True)
        for key in ("0", "1"):
            assert relerr(out[key][0][b], o["cost_pred"]) < TIGHT, (key, b)
            assert relerr(out[key][1][b], o["U_alpha"]) < TIGHT, (key, b)
    assert relerr(out["1"][0], out["0"][0]) < 1e-12 and np.array_equal(out["1"][1], out["0"][1])
    assert np.array_equal(out[None][0], out["1"][0])


def _full_batch_parity(*argv):
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("KPILQR_FUSED")}      # the DEFAULT dispatch is what is tested
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "full_batch_parity.py")] + [str(a) for a in argv], cwd=root,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PARITY ")]
    assert line and "within 1e-9 of the oracle in every leg" in r.stdout, r.stdout[-2000:]
    return json.loads(line[-1][len("PARITY "):])


def test_headline_dispatch_one_wave_raw_uniform_at_batch_1024():
    """The bench's kernels at the size where they are the library's DEFAULT (round-3 verdict, Weak 2): 1024 distinct seeds,
    key-point ordered payload, default environment -> `k_backward_fused_excl<14,7,RU0,RAW,UNI>` ("...:w1:raw:uni:ru0", read back
    through kpilqr_last_launch and asserted by the tool), 64 trajectories spread over the batch against the oracle at 1e-9
    (iLQR.cpp:535-634 with the differencing of Differentiator.cpp:166-222,441-457 inside), and ALL 1024 trajectories' K, k,
    delta_J, costs bit-identical with the `KPILQR_FUSED_RAW=0` leg (differencing kernel + plain sweep) and within 1e-12 of the
    constant residual-Jacobian leg (":rxc": Lzz from the resident r_x' W r_x tile, one product instead of four).  A process of its own: the oracle workers fork before anything touches the GPU."""
    s = _full_batch_parity(1024, 1000, "--sample", 64)
    legs = s["legs"]
    assert legs["A"]["backward"].endswith(":w1:raw:uni:ru0") and legs["A"]["forward"].endswith(":w1:uni:ru0"), legs["A"]
    assert legs["B"]["backward"].endswith(":w1:kpc:uni:ru0") and legs["B"]["bit_identical"]
    assert legs["C"]["backward"].endswith(":rxc") and legs["C"]["forward"].endswith(":rxc") and legs["C"]["agrees_1e12"]
    assert s["checked"] == 64 and max(l["max_rel_err_K"] for l in legs.values()) < 1e-9


def test_bench_form_at_the_bench_size():
    """The bench line's exact kernels at the bench's exact size (round-4 verdict, Weak 2): 1024 DISTINCT seeds x T = 3000, default
    environment, leg C = `...:w1:raw:uni:ru0:rxc | ...:w1:uni:ru0:rxc` (ASSERTED), 16 trajectories spread over the batch against the
    oracle at 1e-9 (its rows computed by forked workers before the GPU is touched), all 1024 within 1e-12 of the per-step-Jacobian
    leg A, which is bit-identical with the differencing-kernel leg B."""
    s = _full_batch_parity(1024, 3000, "--sample", 16)
    legs = s["legs"]
    assert s["batch"] == 1024 and s["T"] == 3000 and s["checked"] == 16
    assert legs["C"]["backward"].endswith(":w1:raw:uni:ru0:rxc") and legs["C"]["forward"].endswith(":w1:uni:ru0:rxc"), legs["C"]
    assert legs["A"]["backward"].endswith(":w1:raw:uni:ru0") and legs["B"]["bit_identical"] and legs["C"]["agrees_1e12"]
    assert max(l["max_rel_err_K"] for l in legs.values()) < 1e-9 and max(l["max_rel_err_cost_pred"] for l in legs.values()) < 1e-9


def test_headline_dispatch_helper_pair_at_batch_320():
    """B <= 512 (a GPU's share of 1024 on two or more GPUs): the consumer / helper pair whose helper wave differences the
    payload (":pairh:raw:uni:ru0"), against the oracle, bit-identical with the differencing kernel in front of it and with the
    constant residual Jacobian (":rxc")."""
    s = _full_batch_parity(320, 1000, "--sample", 64)
    legs = s["legs"]
    assert legs["A"]["backward"].endswith(":pairh:raw:uni:ru0") and legs["B"]["backward"].endswith(":pairh:kpc:uni:ru0")
    assert legs["C"]["backward"].endswith(":pairh:raw:uni:ru0:rxc")
    # forward: the state / cost wave pair of uniform key-point sets (the state wave interpolates its own operands)
    assert legs["A"]["forward"].endswith(":pair:uni:ru0") and legs["C"]["forward"].endswith(":pair:uni:ru0:rxc"), (legs["A"], legs["C"])
    assert legs["B"]["bit_identical"] and legs["C"]["agrees_1e12"]


def test_every_trajectory_of_a_distinct_seed_batch_matches_the_oracle():
    """B = 256 (the consumer / helper pair, ":pairh:raw:uni:ru0") and, as a leg that really is another kernel, the one-wave raw
    sweep forced on the same batch: K, k, predicted costs, delta_J and status of EVERY trajectory against the oracle.  (The committed 1024 x 3000 run: profiles/r04_full_batch_parity.txt.)"""
    s = _full_batch_parity(256, 1000)
    legs = s["legs"]
    assert s["checked"] == 256
    assert legs["A"]["backward"].endswith(":pairh:raw:uni:ru0") and legs["B"]["backward"].endswith(":pairh:kpc:uni:ru0")
    assert legs["E"]["backward"].endswith(":w1:raw:uni:ru0")
