"""Backward wave forms on a task WITH control residuals (r_u != 0: no RU0 / RXC instantiations), GPU box:
python tools/pair_forms_dense.py [B ...]   -- KPILQR_FUSED_WAVES = 3 (pair), 4 (triple, B <= 256), 5 (consumer / helper pair)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from trajoptkp_amd import Engine, synth
T = 3000
p0 = synth.make_problem(task="panda_reaching", T=T, batch=8, min_N=5, dense_residuals=True)
for B in [int(a) for a in sys.argv[1:]] or [512, 256]:
    p = synth.tile_problem(p0, B // 8)
    for waves in ("3", "4", "5"):
        if waves == "4" and B > 256: continue
        os.environ["KPILQR_FUSED_WAVES"] = waves
        with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
            synth.upload(e, p, kp_ordered=True)
            lam = np.full(B, p["lam"])
            e.backward(lam, 100, fetch=False); e.sync()
            raw = ":raw" in e.last_launch("backward")
            ts = []
            for _ in range(8):
                t0 = time.perf_counter()
                if not raw: e.fd_difference()
                e.backward(None, 100, fetch=False); e.sync()
                ts.append(time.perf_counter() - t0)
            print(f"B={B} waves={waves} {e.last_launch('backward')}: {1e3 * min(ts):.3f} ms (differencing {'inside' if raw else 'by its kernel, included'})", flush=True)
