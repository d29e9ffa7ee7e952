"""Diagnostic: what makes the backward kernel slower right after a kernel that wrote the records?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajoptkp_amd import Engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
uniq = 8
p = synth.tile_problem(synth.make_problem(T=T, batch=uniq), B // uniq)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
e = Engine(p["dof"], p["m"], T, p["nr"], batch=B, stream=stream.cuda_stream)
synth.upload(e, p)
e.backward(np.full(B, 0.1), 100, fetch=False); e.forward_linear(np.array([(i / 6.0) ** 2 for i in range(1, 7)]), fetch=False)
e.iterate(); e.sync()
bwd = lambda: e.backward(None, 100, fetch=False)
def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); fn(); b.record(stream); b.synchronize()
    return a.elapsed_time(b)
def case(name, pre):
    pre(); print(f"B={B} T={T} bwd after {name:28s}: {timed(bwd):7.3f} ms", flush=True)
case("bwd", bwd)
case("cost", e.cost_derivs)
case("cost + sync + sleep 30ms", lambda: (e.cost_derivs(), e.sync(), time.sleep(0.03)))
case("cost + traj_cost kernel", lambda: (e.cost_derivs(), e._ck(e._L.kpilqr_trajectory_cost(e._h, None))))
case("cost + fwd", lambda: (e.cost_derivs(), e.forward_linear(None, fetch=False)))
case("interp", e.interpolate)
case("fd only", e.fd_difference)
case("bwd", bwd)
