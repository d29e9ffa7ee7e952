"""Resident-input iteration throughput with the batch cut into trajectory chunks on the library's pipeline streams
(kpilqr_iterate_streamed without uploads / downloads): one chunk's fd_difference overlaps the other chunks' sweeps.
python tools/chunk_overlap.py [batch] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth
from oracle import oracle as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = 3000
p = synth.tile_problem(synth.make_problem(task="panda_reaching", T=T, batch=16, min_N=5), B // 16)
with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
    synth.upload(e, p)
    e.fd_difference(); e.backward(p["lam"], 100, fetch=False)          # lambda and alphas resident from here on
    e.forward_linear(orc.alphas(6), fetch=False); e.sync()
    for mode in ("iterate", 1, 2, 3, "iterate"):       # streamed modes without an FD payload skip fd_difference (records resident)
        def step():
            if mode == "iterate": e.iterate(None, 100, None)
            elif mode == "stages": e.fd_difference(); e.backward(None, 100, fetch=False); e.forward_linear(None, fetch=False)
            else: e.iterate_streamed(nchunks=mode)
        for _ in range(3): step()
        e.sync(); t0 = time.perf_counter()
        for _ in range(steps): step()
        e.sync(); dt = (time.perf_counter() - t0) / steps
        print(f"B={B} {str(mode):8s}: {dt*1e3:7.3f} ms per batch-iteration, {B/dt:9.0f} trajectory-iterations/s", flush=True)
