"""Randomised parity cases shared by tests/test_gpu_fuzz.py (fixed seed, in the -m gpu suite) and tools/fuzz_parity.py (any
seed / count, from the command line): random horizons, batch sizes, key-point intervals or ragged per-DoF key-point lists,
regularisation, PD-check strides, one-sided FD fractions, residuals with and without control Jacobians, over every kernel
family, every wave organisation of the fused sweeps and both forms of the tiled sweeps (a6 inside or not), each compared
with the CPU oracle."""
import os

import numpy as np

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth

TASKS = ["panda_reaching", "acrobot", "hopper", "pentabot", "panda_pushing", "walker", "arm8", "arm5x2", "high_dof_push",
         "quadruped", "humanoid_fixed"]
ENV_KEYS = ("KPILQR_FUSED_WAVES", "KPILQR_FUSED_FWD_WAVES", "KPILQR_TILED_A6", "KPILQR_TILED_FSC")
TOL = 1e-8


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def draw_case(rng, case):
    """Parameters of case number `case` (consumes the generator in a fixed order, so a seed names a sweep)."""
    task = TASKS[case % len(TASKS)]
    big = task in ("high_dof_push", "humanoid_fixed", "quadruped")
    c = dict(case=case, task=task)
    c["T"] = int(rng.choice([2, 3, 5, 17, 64, 129, 300] if big else [2, 3, 5, 17, 64, 129, 300, 777, 1500]))
    c["batch"] = int(rng.integers(1, 4))
    c["min_N"] = int(rng.integers(1, 9))
    c["lam"] = float(10.0 ** rng.uniform(-4, 1))
    c["pd"] = int(rng.choice([1, 7, 100]))
    c["osf"] = float(rng.choice([0.0, 0.1, 0.5]))
    c["fused"] = bool(rng.integers(0, 2))
    c["form"] = str(rng.choice(["auto", "one"]))
    c["a4"], c["a6"] = str(rng.integers(0, 2)), str(rng.choice(["", "0", "1"]))
    c["dense_res"] = bool(rng.integers(0, 2))
    c["ragged"] = bool(c["T"] >= 5 and rng.uniform() < 0.4)
    if c["ragged"]:      # ragged per-DoF lists (bisection-shaped, very different densities per DoF)
        dof = synth.TASKS[task]["dof"]
        c["rows"] = [synth.bisect_keypoints(rng, dof, c["T"], int(rng.integers(1, 4)), rng.uniform(0.0, 1.0, dof)) for _ in range(c["batch"])]
    c["config_id"] = int(rng.integers(1, 6))
    # round 4: payload form (job lists / key-point ordered records / host-differenced columns), differencing explicit or left to
    # the sweeps (the raw forms), ONE constant residual Jacobian where the task has one
    c["payload"] = str(rng.choice(["jobs", "kp_ordered", "columns"]))
    c["explicit_fd"] = bool(rng.integers(0, 2))
    c["rx_const"] = bool(rng.integers(0, 2))
    return c


def run_case(c, worst=None):
    """Runs one case on the GPU and against the oracle; returns the kernel-variant string.  Raises AssertionError."""
    saved = {key: os.environ.get(key) for key in ENV_KEYS}
    try:
        for key in ENV_KEYS:
            os.environ.pop(key, None)
        if c["form"] == "one":
            os.environ["KPILQR_FUSED_WAVES"] = "1"; os.environ["KPILQR_FUSED_FWD_WAVES"] = "1"
        if c["a6"]:
            os.environ["KPILQR_TILED_A6"] = c["a6"]
        T, batch, lam, pd = c["T"], c["batch"], c["lam"], c["pd"]
        if c["ragged"]:
            p = synth.make_ragged_problem(c["task"], T, c["rows"], config_id=c["config_id"], dense_residuals=c["dense_res"],
                                          one_sided_frac=c["osf"], lam=lam)
        else:
            p = synth.make_problem(task=c["task"], T=T, batch=batch, min_N=c["min_N"], dense_residuals=c["dense_res"],
                                   one_sided_frac=c["osf"], lam=lam, config_id=c["config_id"])
        with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=c["fused"]) as e:
            payload = c.get("payload", "jobs")
            synth.upload(e, p, kp_ordered=payload != "jobs", rx_const=c.get("rx_const", False))
            if payload == "columns":
                e.upload_kp_columns(e.kp_columns(*synth.kp_ordered_payload(p), eps=p["eps"]))
            if c.get("explicit_fd", True) or "fused" not in e.backward_variant:
                e.fd_difference()
            if "fused" not in e.backward_variant:
                tail = e.backward_variant.rsplit("_", 1)[-1] if "tiled_" in e.backward_variant else ""
                e.interpolate()
                if "a6" not in tail: e.cost_derivs()
            st, dJ = e.backward(lam, pd)
            K, k = e.gains()
            cost, U = e.forward_linear(orc.alphas(6), want_U=True)
            var = e.last_launch("backward") + " / " + e.last_launch("forward")
    finally:
        for key, val in saved.items():
            if val is None: os.environ.pop(key, None)
            else: os.environ[key] = val
    for b in range(batch):
        o = pipeline.run_trajectory(p, b, lam=lam, pd_stride=pd, want_U=True)
        assert st[b] == o["status"], (c["case"], c["task"], T, st[b], o["status"])
        if o["status"] != 0:
            continue
        errs = dict(K=relerr(K[b], o["K"]), k=relerr(k[b], o["k"]), dJ=abs(dJ[b] - o["delta_J"]) / max(abs(o["delta_J"]), 1e-300),
                    cost=relerr(cost[b], o["cost_pred"]), U=relerr(U[b], o["U_alpha"]))
        if worst is not None:
            w = worst.setdefault(var, {})
            for key, v in errs.items():
                w[key] = max(w.get(key, 0.0), v)
        assert max(errs.values()) < TOL, (c["case"], c["task"], T, batch, c["min_N"], lam, pd, var, errs)
    return var
