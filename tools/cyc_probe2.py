"""Debug probe for a -DKP_CYC build of libkpilqr.so (KPILQR_LIB=...): where the cycles of a step of the headline backward sweep go."""
import sys, os
sys.path.insert(0, ".")
import numpy as np
from trajoptkp_amd import Engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = 3000
p0 = synth.make_problem(task="panda_reaching", T=T, batch=8, min_N=5)
p = synth.tile_problem(p0, B // 8)
with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
    synth.upload(e, p, kp_ordered=True, rx_const=True)
    e.backward(p["lam"], 100, fetch=False); e.sync()
    st, dJ = e.backward(p["lam"], 100)
    print(e.last_launch("backward"))
    K, k = e.gains()
v = K.reshape(B, -1)[:, :8].mean(axis=0)
names = ["peeled steps", "inner steps", "sweep", "a: lerp, a6, requests", "b: a + Tu..Qzz", "c: refresh", "d: gains, V'", "e: LDS symmetrise"]
print(f"crossings {dJ.mean() / T:8.1f} cycles per step")
for nme, x in zip(names, v):
    print(f"{nme:28s} {x / T:8.1f} cycles per step")
