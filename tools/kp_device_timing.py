"""Timing of on-device key-point placement at the headline shape: Panda, T=3000, B=1024."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from trajoptkp_amd import Engine

B, T, dof = 1024, 3000, 7
rng = np.random.default_rng(0)
X = rng.standard_normal((8, T, 2 * dof)).cumsum(axis=1) * 0.01
X = np.tile(X, (B // 8, 1, 1))
with Engine(dof, 7, T, 14, batch=B) as e:
    t0 = time.perf_counter(); e.upload_states(X); e.sync(); up = time.perf_counter() - t0
    for method, thr in (("set_interval", None), ("adaptive_jerk", np.full(dof, 100.0)), ("velocity_change", np.full(dof, 1.0))):
        e.generate_keypoints(method, 5, 100, thr, 0.008); e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.generate_keypoints(method, 5, 100, thr, 0.008)
        e.sync()
        dt = (time.perf_counter() - t0) / 5
        offs, times = e.get_keypoints()
        print(f"{method:16s}: {dt*1e3:7.3f} ms per batch (placement + scan + fill + segmap), {len(times)/(B*dof*T)*100:5.1f} % key-points", flush=True)
    print(f"state upload (pageable, {X.nbytes/1e6:.0f} MB): {up*1e3:.1f} ms")
