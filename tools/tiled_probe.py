"""Diagnostic: tiled MFMA backward vs oracle across state dimensions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajoptkp_amd import Engine, synth
from oracle import pipeline, oracle as orc
for dof in (7, 9, 15):
    synth.TASKS["syn"] = dict(dof=dof, m=7, nr=6, dt=0.008, lim=[87, 87, 87, 87, 12, 12, 12],
                              w_run=[1.0, 0.5, 0.1, 0.1, 0.1, 0.1], w_term=[100.0, 50.0, 1, 1, 1, 1])
    p = synth.make_problem(task="syn", T=20, batch=1, min_N=4, dense_residuals=True)
    o = pipeline.run_trajectory(p, 0)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=1, tiled=True) as e:
        synth.upload(e, p)
        e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, dJ = e.backward(p["lam"])
        K, k = e.gains()
        v = e.backward_variant
    eK = np.max(np.abs(K[0] - o["K"]), axis=(1, 2)) / np.max(np.abs(o["K"]))
    ek = np.max(np.abs(k[0] - o["k"]), axis=1) / np.max(np.abs(o["k"]))
    print(f"dof {dof:2d} n {2*dof:2d} {v:15s} st {st[0]} K err max {eK.max():.2e}  k err by t (T-1 .. 0): " + " ".join(f"{x:.0e}" for x in ek[::-1][:8]), f" dJ rel {abs(dJ[0]-o['delta_J'])/abs(o['delta_J']):.1e}", flush=True)
