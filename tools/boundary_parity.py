"""Parity at the batch sizes where the fused sweeps change their wave organisation (#SIMDs/4 = 256, #SIMDs/2 = 512 on an
MI355X): a few trajectories of each batch against the CPU oracle.  Not part of the test suite (run on the GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth
from oracle import oracle as orc, pipeline


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


for task, T in (("panda_reaching", 300), ("hopper", 200), ("acrobot", 150)):
    p8 = synth.make_problem(task=task, T=T, batch=8, min_N=4, dense_residuals=True, one_sided_frac=0.1)
    ref = [pipeline.run_trajectory(p8, b, want_U=True) for b in range(8)]
    for B in (256, 264, 512, 520, 1024, 1032):
        p = synth.tile_problem(p8, B // 8)
        with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
            synth.upload(e, p)
            e.fd_difference()
            st, dJ = e.backward(p["lam"], 100)
            K, k = e.gains()
            cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        worst = 0.0
        for b in list(range(8)) + [B // 2 + 3, B - 1]:
            o = ref[b % 8]
            assert st[b] == 0
            worst = max(worst, relerr(K[b], o["K"]), relerr(k[b], o["k"]), relerr(cost[b], o["cost_pred"]), relerr(U[b], o["U_alpha"]),
                        abs(dJ[b] - o["delta_J"]) / abs(o["delta_J"]))
        assert worst < 1e-9, (task, B, worst)
        print(f"{task:15s} B={B:5d}: worst relative error {worst:.1e}  ok", flush=True)
