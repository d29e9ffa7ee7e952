"""Diagnostic: run [cost_derivs, backward] x3 then [backward] x3 so that rocprofv3 --pmc shows the
backward kernel's counters right after a writer kernel (cold) and right after itself (warm)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajoptkp_amd import Engine, synth
B, T, uniq = 1024, 3000, 8
p = synth.tile_problem(synth.make_problem(T=T, batch=uniq), B // uniq)
e = Engine(p["dof"], p["m"], T, p["nr"], batch=B)
synth.upload(e, p)
e.backward(np.full(B, 0.1), 100, fetch=False); e.forward_linear(np.array([(i / 6.0) ** 2 for i in range(1, 7)]), fetch=False)
e.iterate(); e.sync()
for _ in range(3):
    e.cost_derivs(); e.backward(None, 100, fetch=False)
e.sync()
for _ in range(3):
    e.backward(None, 100, fetch=False)
e.sync()
print("done")
