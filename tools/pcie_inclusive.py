"""PCIe-inclusive iteration rate: every iteration re-uploads the FD payload and the
residuals + Jacobians from PINNED host memory and downloads K,k, as a host that re-linearises every
iteration would.  Not the bench metric (bench.py keeps inputs resident)."""
import sys, time, ctypes
import numpy as np
import torch
sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth
from trajoptkp_amd.engine import _ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, steps = 3000, 5
p0 = synth.make_problem(task="panda_reaching", T=T, batch=8, min_N=5)
p = synth.tile_problem(p0, B // 8)


def pinned(a):
    t = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)
    v = t.numpy(); v[...] = a
    return t, v

keep = []
for k in ("xplus", "xminus", "xnom", "r", "r_x", "r_u"):
    t, v = pinned(p[k]); keep.append(t); p[k] = v
e = Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True)
synth.upload(e, p)
tK, K = pinned(np.zeros((B, T, e.n, e.m))); tk, kk = pinned(np.zeros((B, T, e.m)))
lam = np.full(B, 0.1)
up_bytes = sum(p[k].nbytes for k in ("xplus", "xminus", "xnom", "r", "r_x", "r_u"))
dn_bytes = K.nbytes + kk.nbytes


def one(upload=True):
    if upload:
        e.upload_fd(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"],
                    job_nom=p["job_nom"], xnom=p["xnom"], eps=p["eps"])
        if upload == "analytic":          # closed-form residual Jacobians stay resident (reaching: selector rows)
            e.upload_residuals(p["r"])
        else:
            e.upload_residuals(p["r"], p["r_x"], p["r_u"])
    e.iterate(lam)
    if upload:
        e._ck(e._L.kpilqr_download_gains(e._h, _ptr(K), _ptr(kk)))
    e.sync()

for mode in (True, "analytic", False):
    if mode == "analytic":
        up_bytes = sum(p[k].nbytes for k in ("xplus", "xminus", "xnom", "r"))
    one(mode); one(mode)
    t0 = time.perf_counter()
    for _ in range(steps):
        one(mode)
    dt = (time.perf_counter() - t0) / steps
    label = {True: "PCIe-inclusive          ", "analytic": "PCIe-inclusive, J const ", False: "resident                "}[mode]
    print(f"B={B} {label}: {dt*1e3:8.2f} ms/batch-iteration = {B/dt:9.1f} trajectory-iterations/s"
          + (f"   (H2D {up_bytes/1e9:.2f} GB + D2H {dn_bytes/1e9:.2f} GB per iteration -> {(up_bytes+dn_bytes)/dt/1e9:.1f} GB/s over PCIe)" if mode else ""), flush=True)
