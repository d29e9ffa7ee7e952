"""Diagnostic: time each engine stage in isolation (repeated back to back) and in pipeline order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajoptkp_amd import Engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
uniq = 8
p = synth.make_problem(T=T, batch=uniq)
p = synth.tile_problem(p, B // uniq)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
e = Engine(p["dof"], p["m"], T, p["nr"], batch=B, stream=stream.cuda_stream)
synth.upload(e, p)
e.backward(np.full(B, 0.1), 100, fetch=False); e.forward_linear(np.array([(i / 6.0) ** 2 for i in range(1, 7)]), fetch=False)
e.iterate(); e.sync()
st = {"fd": e.fd_difference, "interp": e.interpolate, "cost": e.cost_derivs,
      "bwd": lambda: e.backward(None, 100, fetch=False), "fwd": lambda: e.forward_linear(None, fetch=False)}
def timeit(fn, reps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): fn()
    b.record(stream); b.synchronize()
    return a.elapsed_time(b) / reps
for k, fn in st.items():
    print(f"{k:7s} alone x5 : {timeit(fn):8.3f} ms")
print("pipeline order, per stage:")
for k, fn in st.items():
    print(f"{k:7s} single  : {timeit(fn, 1):8.3f} ms")
print("bwd after cost:", end=" "); e.cost_derivs(); print(f"{timeit(st['bwd'], 1):8.3f} ms")
print("bwd after bwd :", end=" "); print(f"{timeit(st['bwd'], 1):8.3f} ms")
res = e.results(); print("status ok:", int((res['status'] == 0).sum()), "dJ0", res['delta_J'][0])
