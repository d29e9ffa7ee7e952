"""Refresh histogram of the running inverse (kpilqr_backward_stats) on the bench's key-point workloads (GPU box):
python tools/refresh_hist.py [batch]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
from trajoptkp_amd import Engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
names = ("third_order_only", "plus_1_second_order", "plus_2_second_order", "plus_3_second_order", "ldl_factorisation", "pivoted_slow_path")
for kind in ("set_interval", "reach_velocity_change", "reach_adaptive_jerk", "adaptive_jerk", "iterative_error"):
    p, p0, desc = bench.build_problem(kind, B, 3000, 5, "panda_reaching", distinct=False)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=B, fused=True) as e:
        synth.upload(e, p, kp_ordered=True, rx_const=True)
        st, _ = e.backward(np.full(B, p["lam"]), 100)
        h = e.backward_stats(100)
    tot = h[st == 0].sum(0).astype(float)
    print(f"{kind:24s} " + "  ".join(f"{n}={v / tot.sum():.3f}" for n, v in zip(names, tot)), flush=True)
