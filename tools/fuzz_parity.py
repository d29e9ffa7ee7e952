"""Randomised parity sweep on the GPU box (not part of the test suite): random horizons, batch sizes, key-point
intervals or ragged per-DoF key-point lists, regularisation, PD-check strides, one-sided FD fractions, residuals with and
without control Jacobians, over every kernel family and every fusion form of the tiled sweeps (a4 / a6), each compared with
the CPU oracle.  Usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth
from oracle import oracle as orc, pipeline

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
TASKS = ["panda_reaching", "acrobot", "hopper", "pentabot", "panda_pushing", "walker", "arm8", "arm5x2", "high_dof_push"]


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


worst = {}
t0 = time.time()
for case in range(N):
    task = TASKS[case % len(TASKS)]
    big = task == "high_dof_push"
    T = int(rng.choice([2, 3, 5, 17, 64, 129, 300] if big else [2, 3, 5, 17, 64, 129, 300, 777, 1500]))
    batch = int(rng.integers(1, 4))
    min_N = int(rng.integers(1, 9))
    lam = float(10.0 ** rng.uniform(-4, 1))
    pd = int(rng.choice([1, 7, 100]))
    osf = float(rng.choice([0.0, 0.1, 0.5]))
    fused = bool(rng.integers(0, 2))
    form = str(rng.choice(["auto", "one"]))
    for k in ("KPILQR_FUSED_WAVES", "KPILQR_FUSED_FWD_WAVES", "KPILQR_TILED_A4", "KPILQR_TILED_A6"):
        os.environ.pop(k, None)
    if form == "one":
        os.environ["KPILQR_FUSED_WAVES"] = "1"; os.environ["KPILQR_FUSED_FWD_WAVES"] = "1"
    a4, a6 = str(rng.integers(0, 2)), str(rng.choice(["", "0", "1"]))
    os.environ["KPILQR_TILED_A4"] = a4
    if a6:
        os.environ["KPILQR_TILED_A6"] = a6
    dense_res = bool(rng.integers(0, 2))
    if T >= 5 and rng.uniform() < 0.4:          # ragged per-DoF lists (bisection-shaped, very different densities per DoF)
        dof = synth.TASKS[task]["dof"]
        rows = [synth.bisect_keypoints(rng, dof, T, int(rng.integers(1, 4)), rng.uniform(0.0, 1.0, dof)) for _ in range(batch)]
        p = synth.make_ragged_problem(task, T, rows, config_id=int(rng.integers(1, 6)), dense_residuals=dense_res, one_sided_frac=osf, lam=lam)
        min_N = -1
    else:
        p = synth.make_problem(task=task, T=T, batch=batch, min_N=min_N, dense_residuals=dense_res,
                               one_sided_frac=osf, lam=lam, config_id=int(rng.integers(1, 6)))
    with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=fused) as e:
        synth.upload(e, p)
        e.fd_difference()
        if "fused" not in e.backward_variant:
            tail = e.backward_variant.rsplit("_", 1)[-1] if "tiled_" in e.backward_variant else ""
            if "a4" not in tail: e.interpolate()
            if "a6" not in tail: e.cost_derivs()
        st, dJ = e.backward(lam, pd)
        K, k = e.gains()
        cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        var = e.backward_variant + "/" + e.forward_variant + ("/" + form if "fused" in e.backward_variant else "")
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=pd, want_U=True)
        assert st[b] == o["status"], (case, task, T, st[b], o["status"])
        if o["status"] != 0:
            continue
        errs = dict(K=relerr(K[b], o["K"]), k=relerr(k[b], o["k"]), dJ=abs(dJ[b] - o["delta_J"]) / max(abs(o["delta_J"]), 1e-300),
                    cost=relerr(cost[b], o["cost_pred"]), U=relerr(U[b], o["U_alpha"]))
        for key, v in errs.items():
            w = worst.setdefault(var, {})
            w[key] = max(w.get(key, 0.0), v)
        assert max(errs.values()) < 1e-8, (case, task, T, batch, min_N, lam, pd, var, errs)
    print(f"case {case:3d} {task:15s} T={T:5d} B={batch} min_N={min_N} lam={lam:.2e} pd={pd:3d} {var}  ok", flush=True)
print(f"{N} cases in {time.time() - t0:.1f} s; worst relative errors per kernel family:")
for var, w in sorted(worst.items()):
    print(f"  {var:55s} " + " ".join(f"{k}={v:.1e}" for k, v in w.items()))
