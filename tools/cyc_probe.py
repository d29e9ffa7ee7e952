"""Debug probe for a -DKP_CYC build of libkpilqr.so (KPILQR_LIB=...): cycles of the UNI backward sweep's crossings."""
import sys, os
sys.path.insert(0, ".")
import numpy as np
from trajoptkp_amd import Engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for mn in (1, 5, 100):
    p0 = synth.make_problem(task="panda_reaching", T=3000, batch=8, min_N=mn)
    p = synth.tile_problem(p0, B // 8)
    for raw in (True, False):
        os.environ["KPILQR_FUSED_RAW"] = "1" if raw else "0"
        with Engine(p["dof"], p["m"], 3000, p["nr"], batch=B, fused=True) as e:
            synth.upload(e, p, kp_ordered=True)
            e.backward(p["lam"], 100, fetch=False); e.sync()
            st, dJ = e.backward(p["lam"], 100)
            K, k = e.gains()
        ncross = len(np.nonzero(np.diff(p0["kp_rows"][0][0]))[0])
        peel, inner, tot, pa = (K[:, 0].reshape(B, -1)[:, i].mean() for i in range(4))
        print(f"min_N {mn:3d} raw {int(raw)}: per crossing {dJ.mean() / ncross:7.1f} | peeled step {peel / ncross:7.1f} | inner step {inner / max(3000 - ncross, 1):7.1f} | "
              f"phase a (lerp, a6, requests) per step {pa / 3000:7.1f} | total per step {tot / 3000:7.1f} cycles")
