"""Quick parity check of the fused sweeps for one library build (GPU box): python tools/ab_check.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth
from oracle import oracle as orc, pipeline
worst = 0.0
for task, T, B, dense, minN in (("panda_reaching", 300, 3, False, 5), ("panda_reaching", 257, 2, True, 3), ("acrobot", 200, 2, True, 4), ("hopper", 150, 2, True, 7)):
    p = synth.make_problem(task=task, T=T, batch=B, min_N=minN, dense_residuals=dense, one_sided_frac=0.1)
    with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
        synth.upload(e, p)
        e.fd_difference()
        st, dJ = e.backward(p["lam"], 100)
        K, k = e.gains()
        cost = e.forward_linear(orc.alphas(6))
        var = e.backward_variant
    for b in range(B):
        o = pipeline.run_trajectory(p, b)
        err = max(np.abs(K[b] - o["K"]).max() / np.abs(o["K"]).max(), np.abs(k[b] - o["k"]).max() / np.abs(o["k"]).max(),
                  abs(dJ[b] - o["delta_J"]) / abs(o["delta_J"]), np.abs(cost[b] - o["cost_pred"]).max() / np.abs(o["cost_pred"]).max())
        worst = max(worst, err)
        assert st[b] == o["status"] and err < 1e-9, (task, b, err)
print("parity ok", var, "worst rel err %.2e" % worst)
