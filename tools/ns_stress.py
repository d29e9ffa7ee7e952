"""Ad-hoc numerical stress of the running-inverse fast path: long horizon, small and large lambda, dense residuals,
against the CPU oracle (test infrastructure).  Prints the worst relative error of K and k per case."""
import sys; sys.path.insert(0, ".")
import numpy as np
from oracle import pipeline
from trajoptkp_amd import Engine, synth

def relerr(a, b): return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))
for task, T, kw in (("panda_reaching", 3000, {}), ("panda_reaching", 1500, dict(dense_residuals=True)), ("acrobot", 800, dict(config_id=1, dense_residuals=True))):
    for lam in (1e-4, 1e-2, 10.0):
        p = synth.make_problem(task=task, T=T, batch=2, min_N=5, **kw)
        for fused in (True, False):
            with Engine(p["dof"], p["m"], T, p["nr"], batch=2, fused=fused) as e:
                synth.upload(e, p)
                e.fd_difference()
                if not fused: e.interpolate(); e.cost_derivs()
                st, dJ = e.backward(lam, 100)
                K, k = e.gains()
            worst = 0.0
            for b in range(2):
                o = pipeline.run_trajectory(p, b, lam=lam, stages=("fd", "interp", "cost", "bwd"))
                assert o["status"] == st[b], (o["status"], st[b])
                if st[b] == 0: worst = max(worst, relerr(K[b], o["K"]), relerr(k[b], o["k"]))
            print(f"{task:15s} T={T:5d} lam={lam:7.0e} {'fused' if fused else 'mater'}: status {list(st)} worst rel err {worst:.2e}", flush=True)
