"""Wide-control tiled sweeps (tiled_wide.hip) against the generic VALU / LDS kernels on the humanoid shape (GPU box):
python tools/wide_timing.py [batch] [T]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from trajoptkp_amd import Engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
for task in ("humanoid", "humanoid_fixed", "quadruped"):
    p0 = synth.make_problem(task=task, T=T, batch=2, min_N=5)
    p = synth.tile_problem(p0, B // 2)
    for generic in (False, True):
        with Engine(p["dof"], p["m"], T, p["nr"], batch=B, generic=generic) as e:
            synth.upload(e, p)
            e.fd_difference(); e.interpolate(); e.cost_derivs()
            alphas = np.array([(i / 6.0) ** 2 for i in range(1, 7)])
            e.backward(p["lam"], 100, fetch=False); e.forward_linear(alphas, fetch=False); e.sync()
            reps = 3 if not generic else 1
            t0 = time.perf_counter()
            for _ in range(reps): e.backward(None, 100, fetch=False)
            e.sync(); tb = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps): e.forward_linear(None, fetch=False)
            e.sync(); tf = (time.perf_counter() - t0) / reps
            print(f"{task:15s} n={p['n']} m={p['m']} B={B} T={T} {e.backward_variant:14s}/{e.forward_variant:14s}: backward {1e3 * tb:8.2f} ms  forward {1e3 * tf:8.2f} ms", flush=True)
