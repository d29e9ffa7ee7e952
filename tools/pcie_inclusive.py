"""PCIe-inclusive iteration rate: every iteration re-uploads the FD payload and the residuals
(+ Jacobians) from PINNED host memory and downloads K, k, as a host that re-linearises every iteration does.
Not the bench metric (bench.py keeps inputs resident; it reports these figures in `pcie_inclusive`).

  python tools/pcie_inclusive.py [batch] [steps]

Forms: "serial"  = kpilqr_upload_fd + kpilqr_upload_residuals + kpilqr_iterate + kpilqr_download_gains + sync per iteration
       "chunks=N" = kpilqr_iterate_streamed over N trajectory chunks (H2D | kernels | D2H overlapped; the FD payload goes up
       key-point ordered, kpilqr_fd_kp_layout), one sync at the end
       of the timed loop ("pipelined") or after every iteration ("per-iteration sync").
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from trajoptkp_amd import Engine, synth   # noqa: E402


def measure(B=256, steps=5, T=3000, task="panda_reaching", chunk_list=(4, 8, 16), quiet=False):
    uniq = min(8, B)
    p0 = synth.make_problem(task=task, T=T, batch=uniq, min_N=5)
    p = synth.tile_problem(p0, B // uniq)
    B = p["batch"]
    out = {"batch": B, "T": T, "task": task, "steps": steps}
    with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
        synth.upload(e, p)
        alphas = np.array([(i / 6.0) ** 2 for i in range(1, 7)])
        e.iterate(p["lam"], 100, alphas); e.sync()
        K0, k0 = e.gains()
        pin = {}
        for name in ("r", "r_x", "r_u", "u_nom", "xplus", "xminus", "xnom"):
            pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
        ints = {}
        for name, dt in (("job_b", np.int32), ("job_t", np.int32), ("job_col", np.int32), ("job_nom", np.int32), ("job_mode", np.uint8)):
            ints[name] = e.pinned(p[name].shape, dt); ints[name][...] = p[name]
        lam = e.pinned(B); lam[:] = p["lam"]
        K = e.pinned(K0.shape); k = e.pinned(k0.shape)
        # the chunk pipeline ships the key-point ordered payload (no job lists, no nominal rows; the fused sweeps read it
        # directly); the serial form keeps round 1's call sequence with the job arrays
        kp_payload = synth.kp_ordered_payload(p)
        slab = e.fd_kp_slab(*kp_payload)
        fd_bytes = slab["layout"].bytes
        # ... or the key-point columns a host has differenced itself (kpilqr_upload_kp_columns): 3n doubles per entry
        cols = e.kp_columns(*kp_payload, eps=p["eps"])
        col_bytes = cols["entries"] * 3 * p["n"] * 8
        res_bytes = {True: pin["r"].nbytes + pin["r_x"].nbytes + pin["r_u"].nbytes, False: pin["r"].nbytes}
        dn_bytes = K.nbytes + k.nbytes
        from trajoptkp_amd.engine import _ptr

        def serial(full):
            e.upload_fd(ints["job_b"], ints["job_t"], ints["job_col"], ints["job_mode"], pin["xplus"], pin["xminus"],
                        job_nom=ints["job_nom"], xnom=pin["xnom"], eps=p["eps"])
            if full:
                e.upload_residuals(pin["r"], pin["r_x"], pin["r_u"])
            else:
                e.upload_residuals(pin["r"])
            e.iterate(lam)
            e._ck(e._L.kpilqr_download_gains(e._h, _ptr(K), _ptr(k)))
            e.sync()

        def streamed(full, nchunks, sync_each, columns=False):
            kw = dict(r=pin["r"], r_x=pin["r_x"], r_u=pin["r_u"]) if full else dict(r=pin["r"])
            kw.update(dict(kp_cols=cols) if columns else dict(fd_kp=slab))
            e.iterate_streamed(eps=p["eps"], lam=lam, K=K, k=k, nchunks=nchunks, **kw)
            if sync_each:
                e.sync()

        def timed(fn, *a):
            fn(*a); fn(*a); e.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(*a)
            e.sync()
            return (time.perf_counter() - t0) / steps

        rows = []
        for full in (True, False):
            up = fd_bytes + res_bytes[full]
            label = "full payload" if full else "resident Jacobians"
            dt = timed(serial, full)
            rows.append((label, "serial", dt, up))
            for nc in chunk_list:
                for sync_each in (True, False):
                    K[...] = 0
                    dt = timed(streamed, full, nc, sync_each)
                    assert np.array_equal(K, K0) and np.array_equal(k, k0), "streamed K differs from the staged path"
                    rows.append((label, f"chunks={nc} " + ("per-iteration sync" if sync_each else "pipelined"), dt, up))
        # the host-differenced columns as the payload (pipelined form, every chunk count)
        for full in (True, False):
            for nc in chunk_list:
                K[...] = 0
                dt = timed(streamed, full, nc, False, True)
                assert np.array_equal(K, K0) and np.array_equal(k, k0), "streamed K differs from the staged path (key-point columns)"
                rows.append(("columns + " + ("full residual payload" if full else "resident Jacobians"), f"chunks={nc} pipelined", dt,
                             col_bytes + res_bytes[full]))
        e.iterate(lam); e.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.iterate(lam)
        e.sync()
        dt_res = (time.perf_counter() - t0) / steps
        # Reaching has ONE residual Jacobian (Reaching.cpp:43-54): a host with analytic residuals uploads it once
        # (kpilqr_upload_residual_jacobians_const) and ships, per iteration, the FD payload + the residuals -- the FULL payload
        # of such a task.  Same H2D bytes as "resident Jacobians"; the sweeps no longer read r_x from HBM either.
        if p.get("rx_const") is not None:
            e.upload_residual_jacobians_const(p["rx_const"], None)
            for columns, label in ((False, "constant Jacobians (full payload of a task with analytic residuals)"),
                                   (True, "columns + constant Jacobians")):
                for nc in chunk_list:
                    K[...] = 0
                    dt = timed(streamed, False, nc, False, columns)
                    # (the :rxc sweeps form l_x from the resident r_x' W r_x tile in another accumulation order: K in the same bits, k to 1e-12)
                    assert np.array_equal(K, K0) and np.max(np.abs(k - k0)) <= 1e-12 * np.max(np.abs(k0)), "streamed K differs from the staged path (constant Jacobians)"
                    rows.append((label, f"chunks={nc} pipelined", dt, (col_bytes if columns else fd_bytes) + res_bytes[False]))
            e.iterate(lam); e.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                e.iterate(lam)
            e.sync()
            out["resident_constant_jacobians_traj_it_per_s"] = B / ((time.perf_counter() - t0) / steps)
    out["resident_traj_it_per_s"] = B / dt_res
    out["rows"] = [dict(payload=a, form=b, ms_per_iteration=1e3 * dt, traj_it_per_s=B / dt, h2d_GB=up / 1e9, d2h_GB=dn_bytes / 1e9,
                        link_GBps=(up + dn_bytes) / dt / 1e9) for (a, b, dt, up) in rows]
    if not quiet:
        print(f"B={B} resident: {1e3 * dt_res:8.2f} ms/batch-iteration = {B / dt_res:9.1f} trajectory-iterations/s")
        for r in out["rows"]:
            print(f"B={B} {r['payload']:36s} {r['form']:32s}: {r['ms_per_iteration']:8.2f} ms = {r['traj_it_per_s']:9.1f} traj-it/s"
                  f"  (H2D {r['h2d_GB']:.2f} GB + D2H {r['d2h_GB']:.2f} GB -> {r['link_GBps']:.1f} GB/s)", flush=True)
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    chunks = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (4, 8, 16)
    measure(B, steps, chunk_list=chunks)
