"""GPU tests of the boundary's corner cases (round-2 advisor findings) and of the multi-process C-ABI collective:
r_u written on the device behind the library's back, one-sided FD jobs without nominal rows, streamed iterations whose
slab layout changes between calls, and kpilqr_comm_init / kpilqr_allreduce_linesearch with nranks = 2 from two fresh
processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth
from trajoptkp_amd.engine import KpilqrError

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


@pytest.mark.parametrize("one_wave", [False, True])
def test_r_u_written_through_the_device_pointer(monkeypatch, one_wave):
    """kpilqr_device_ptr(KPILQR_BUF_R_U) hands out a writable pointer: a caller that fills r_u on the device (zero-copy,
    never through kpilqr_upload_residuals) must not get the r_u-free instantiations of the fused sweeps."""
    import torch
    if one_wave:
        monkeypatch.setenv("KPILQR_FUSED_WAVES", "1")
        monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    p = synth.make_problem(task="panda_reaching", T=150, batch=2, min_N=5, dense_residuals=True)
    assert np.any(p["r_u"])
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        e.set_keypoints_rows(p["kp_rows"])
        e.upload_fd(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], job_nom=p["job_nom"], xnom=p["xnom"], eps=p["eps"])
        e.upload_residuals(p["r"], p["r_x"], None, p["w_run"], p["w_term"])          # r_u NOT uploaded
        e.upload_nominal(p["u_nom"], p["ctrl_lim"])
        e.sync()
        dev = torch.as_tensor(e.device_array(5, p["r_u"].shape), device="cuda")      # KPILQR_BUF_R_U
        dev.copy_(torch.from_numpy(p["r_u"]))
        torch.cuda.synchronize()
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results(); K, k = e.gains()
    for b in range(2):
        o = pipeline.run_trajectory(p, b)
        assert res["status"][b] == 0
        assert relerr(K[b], o["K"]) < 1e-9 and relerr(k[b], o["k"]) < 1e-9
        assert relerr(res["cost_pred"][b], o["cost_pred"]) < 1e-9


def test_one_sided_job_without_nominal_rows_is_reported():
    """job_nom = NULL with a one-sided job: the job cannot be differenced (no nominal row); it is skipped on the device and
    reported by the next synchronising call -- kpilqr_sync or a blocking download -- instead of silently using row 0."""
    p = synth.make_problem(task="panda_reaching", T=40, batch=1, min_N=5, one_sided_frac=0.3)
    assert np.any(p["job_mode"] != 0)
    for via in ("sync", "get_AB"):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=1) as e:
            e.set_keypoints_rows(p["kp_rows"])
            e.upload_fd(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], job_nom=None, xnom=p["xnom"], eps=p["eps"])
            e.fd_difference()
            with pytest.raises(KpilqrError) as ei:
                e.sync() if via == "sync" else e.get_AB()
            assert ei.value.code == -1 and "FD job" in str(ei.value)
            e.sync()
    # central jobs never look at the nominal rows: the same call without them is fine
    q = synth.make_problem(task="panda_reaching", T=40, batch=1, min_N=5)
    with Engine(q["dof"], q["m"], q["T"], q["nr"], batch=1) as e:
        e.set_keypoints_rows(q["kp_rows"])
        e.upload_fd(q["job_b"], q["job_t"], q["job_col"], q["job_mode"], q["xplus"], q["xminus"], eps=q["eps"])
        e.fd_difference(); e.sync()


@pytest.mark.parametrize("fused", [True, False])
def test_streamed_iterations_with_a_changing_slab_layout(fused):
    """Consecutive kpilqr_iterate_streamed calls whose job counts / per-trajectory offsets differ (the key-points changed
    between iterations): the second call must not overwrite slab ranges the first one's chunk streams still read.  Each
    result is compared with the ordinary blocking call sequence on the same inputs."""
    T, batch = 160, 9
    probs = [synth.make_problem(task="panda_reaching", T=T, batch=batch, min_N=mn, dense_residuals=True, one_sided_frac=0.2, config_id=cid)
             for mn, cid in ((2, 2), (7, 3), (3, 4))]
    want = []
    for p in probs:
        with Engine(p["dof"], p["m"], T, p["nr"], batch=batch, fused=fused) as e:
            synth.upload(e, p)
            e.iterate(p["lam"], 100, orc.alphas(6))
            K, k = e.gains()
            want.append((K, k, e.results()))
    p0 = probs[0]
    with Engine(p0["dof"], p0["m"], T, p0["nr"], batch=batch, fused=fused) as e:
        e.upload_residuals(None, None, None, p0["w_run"], p0["w_term"])
        e.upload_nominal(None, p0["ctrl_lim"])
        e.set_keypoints_rows(p0["kp_rows"])
        e.forward_linear(orc.alphas(6), fetch=False)              # alphas resident
        outs = []
        for it, p in enumerate(probs):
            e.set_keypoints_rows(p["kp_rows"])
            s = e.fd_slab(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], p["job_nom"], p["xnom"])
            pin = {}
            for name in ("r", "r_x", "r_u", "u_nom"):
                pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
            lam = e.pinned(batch); lam[:] = p["lam"]
            K = e.pinned(want[it][0].shape); k = e.pinned(want[it][1].shape)
            cp = e.pinned((batch, 6)); st = e.pinned(batch, np.int32)
            e.iterate_streamed(fd=s, eps=p["eps"], lam=lam, K=K, k=k, cost_pred=cp, status=st, nchunks=3, **pin)     # no wait in between
            outs.append((K, k, cp, st))
        e.sync()
        outs = [tuple(np.array(a) for a in o) for o in outs]       # the pinned buffers are freed with the engine
    for (K, k, cp, st), (K0, k0, res0) in zip(outs, want):
        assert np.all(st == 0)
        assert np.array_equal(K, K0) and np.array_equal(k, k0)
        assert np.array_equal(cp, res0["cost_pred"])


def test_streamed_offsets_are_validated_before_anything_is_enqueued():
    p = synth.make_problem(task="panda_reaching", T=60, batch=4, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=4, fused=True) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        K0, k0 = e.gains()
        s = e.fd_slab(p["job_b"], p["job_t"], p["job_col"], p["job_mode"], p["xplus"], p["xminus"], p["job_nom"], p["xnom"])
        K = e.pinned(K0.shape)
        good = s["traj_job_first"].copy()
        for bad_at, bad in ((2, int(good[1]) - 1), (3, s["njobs"] + 5), (1, -1)):
            s["traj_job_first"][:] = good
            s["traj_job_first"][bad_at] = bad
            with pytest.raises(KpilqrError) as ei:
                e.iterate_streamed(fd=s, eps=p["eps"], K=K)
            assert ei.value.code == -1
        s["traj_nom_first"][2] = -3
        s["traj_job_first"][:] = good
        with pytest.raises(KpilqrError):
            e.iterate_streamed(fd=s, eps=p["eps"], K=K)
        e.sync()
        K1, _ = e.gains()                                         # the refused calls touched nothing
        assert np.array_equal(K1, K0)


# ---- the C ABI's collective with two ranks -----------------------------------------------------------------------------
_RANK_SCRIPT = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
rank, nranks, idfile, device = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
from trajoptkp_amd import Engine, synth
from trajoptkp_amd.engine import KpilqrError
from trajoptkp_amd import distributed as kd
from oracle import oracle as orc
B = 6
p = synth.make_problem(task="panda_reaching", T=80, batch=B, min_N=5)
lo, hi = kd.shard_range(B, rank, nranks)
q = synth.make_problem(task="panda_reaching", T=80, batch=hi - lo, min_N=5, first_b=lo)
out = dict(rank=rank)
with Engine(q["dof"], q["m"], q["T"], q["nr"], batch=hi - lo, device=device, fused=True) as e:
    synth.upload(e, q)
    e.iterate(q["lam"], 100, orc.alphas(6))
    local = e.allreduce_linesearch()                  # no communicator yet: this rank's sums
    out["local"] = local.tolist()
    if rank == 0:
        uid = e.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f: f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120: raise SystemExit("no unique id")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    try:
        e.comm_init(nranks, rank, uid)
        out["reduced"] = e.allreduce_linesearch().tolist()
    except KpilqrError as ex:
        out["error"] = str(ex)
print("RESULT " + json.dumps(out), flush=True)
"""


def test_device_generated_keypoints_size_the_column_store_and_check_the_payload():
    """Round-3 advisor findings: after kpilqr_generate_keypoints the library knows the number of entries (one int read back), so
    (1) a key-point ordered upload whose `entries` is not that number is refused BEFORE kpilqr_get_keypoints has ever run -- the
    sweeps index the slab by the device lists, a smaller slab would be read past its end -- and (2) the column store is sized
    from the real count, not from the worst case batch * dof * T (Panda, T = 3000, 64 trajectories: 1.35 GB against 90 MB)."""
    import torch
    T, B = 3000, 64
    p0 = synth.make_problem(task="panda_reaching", T=T, batch=1, min_N=5)
    p = synth.tile_problem(p0, B)
    xp, xm, mode = synth.kp_ordered_payload(p)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
        base = free0 - torch.cuda.mem_get_info()[0]               # gains, residuals, Jacobians ...
        e.generate_keypoints("set_interval", 5)
        s = e.fd_kp_slab(xp, xm, mode, pinned=False)
        bad = dict(s); bad["entries"] = s["entries"] - 7
        with pytest.raises(KpilqrError) as ei:
            e.upload_fd_kp(bad)                                     # ... without kpilqr_get_keypoints in between
        assert ei.value.code == -1
        e.upload_fd_kp(s, eps=p["eps"])
        e.upload_residuals(p["r"], p["r_x"], None, p["w_run"], p["w_term"]); e.upload_nominal(p["u_nom"], p["ctrl_lim"])
        e.iterate(p["lam"], 100, orc.alphas(6)); e.sync()
        used = free0 - torch.cuda.mem_get_info()[0] - base
        K, _ = e.gains()
    # key-point times (worst case kept: 5.4 MB), payload slab (~245 MB), column store (~90 MB) -- far from the worst-case 1.35 GB store
    assert used < 600e6, used
    o = pipeline.run_trajectory(p0, 0)
    assert relerr(K[0], o["K"]) < 1e-9 and np.array_equal(K[0], K[B - 1])


def test_streamed_iteration_redifferences_a_resident_job_list_after_new_keypoints():
    """Round-3 advisor finding: kpilqr_iterate_streamed without a new payload, on a fused context whose resident payload is a JOB
    LIST and whose key-points have been set again since (the column store is stale): the chunks have no jobs of their own, so the
    payload is re-differenced on the context first -- the result is kpilqr_iterate's."""
    p = synth.make_problem(task="panda_reaching", T=160, batch=6, min_N=4, dense_residuals=True, one_sided_frac=0.2)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=6, fused=True) as e:
        synth.upload(e, p)
        e.iterate(p["lam"], 100, orc.alphas(6))
        K0, _ = e.gains(); c0 = e.results()["cost_pred"]
        e.set_keypoints_rows(p["kp_rows"])                          # the same lists again: the column store is invalidated
        lam = e.pinned(6); lam[:] = p["lam"]
        K = e.pinned(K0.shape); cp = e.pinned((6, 6))
        e.iterate_streamed(lam=lam, K=K, cost_pred=cp, nchunks=3)
        e.sync()
        assert np.array_equal(K, K0) and np.array_equal(cp, c0)


def test_c_abi_linesearch_allreduce_two_processes(tmp_path):
    """Two fresh processes, one kpilqr_ctx each, kpilqr_comm_unique_id -> kpilqr_comm_init(nranks = 2) ->
    kpilqr_allreduce_linesearch, against the sum of the two ranks' local vectors.  With two GPUs visible the ranks use
    devices 0 and 1 and the reduction must be exact.  On a ONE-GPU box RCCL refuses a communicator whose ranks share a
    device ("Duplicate GPU detected"): then the test checks what can be checked there -- the two-rank rendezvous over the
    distributed id runs, both ranks get RCCL's refusal as a clean KPILQR error (no hang, no crash), and the contexts stay
    usable for the single-rank reduction."""
    import torch
    ngpu = torch.cuda.device_count()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    idfile = str(tmp_path / "rccl_id.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", idfile, str(r if ngpu >= 2 else 0)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for pr in procs:
        try:
            so, se = pr.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL rendezvous hung")
        assert pr.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[7:]))
    outs.sort(key=lambda o: o["rank"])
    total = np.asarray(outs[0]["local"]) + np.asarray(outs[1]["local"])
    assert total[7] == 6                                           # every trajectory had a valid backward pass
    if ngpu >= 2:
        for o in outs:
            assert "error" not in o, o
            assert np.allclose(o["reduced"], total, rtol=1e-14, atol=0.0)
    else:
        for o in outs:
            assert "reduced" not in o and "RCCL" in o["error"], o
