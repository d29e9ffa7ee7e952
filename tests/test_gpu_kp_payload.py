"""GPU tests of the record-free fused path (round 3): the key-point column store, the key-point ordered FD payload
(kpilqr_upload_fd_kp) and the RAW backward sweep that differences it at the segment crossings.  Every form must give the
bytes of the job-list payload + kpilqr_fd_difference path (same arithmetic: Differentiator.cpp:166-222,441-457), and those
are held to the oracle by tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import pipeline
from trajoptkp_amd import Engine, synth
from trajoptkp_amd.engine import KpilqrError

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def _run(p, fused, kp_ordered, explicit_fd=False, lam=None, pd=100, want_AB=False):
    lam = p["lam"] if lam is None else lam
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=fused) as e:
        synth.upload(e, p, kp_ordered=kp_ordered)
        if explicit_fd:
            e.fd_difference()
            st, dJ = e.backward(lam, pd)
            cost, U = e.forward_linear(orc.alphas(6), want_U=True)
        else:
            e.iterate(lam, pd, orc.alphas(6))
            res = e.results(); st, dJ, cost, U = res["status"], res["delta_J"], res["cost_pred"], None
        K, k = e.gains()
        out = dict(K=K, k=k, status=st, delta_J=dJ, cost=cost, U=U, variant=e.backward_variant)
        if want_AB:
            e.interpolate()
            out["A"], out["B"] = e.get_AB()
    return out


def _same(a, b, keys=("K", "k", "delta_J", "cost"), rtol=0.0):
    """rtol = 0: bit for bit; else max |a - b| <= rtol * max |b| per array."""
    for key in keys:
        if rtol == 0.0:
            assert np.array_equal(a[key], b[key]), key
        else:
            u, v = np.asarray(a[key], float), np.asarray(b[key], float)
            assert np.max(np.abs(u - v)) <= rtol * max(float(np.max(np.abs(v))), 1e-300), (key, float(np.max(np.abs(u - v))), float(np.max(np.abs(v))))


@pytest.fixture(params=["one_wave", "auto", "auto_roles"])
def waves(request, monkeypatch):
    """one_wave: the raw backward sweep; auto: small batches -> the consumer / helper pair, whose helper wave differences the
    payload; auto_roles: the same with the two roles alternating with the block index."""
    if request.param == "one_wave":
        monkeypatch.setenv("KPILQR_FUSED_WAVES", "1")
        monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    elif request.param == "auto_roles":
        monkeypatch.setenv("KPILQR_ROLE_SHIFT", "0")
        return "auto"
    return request.param


@pytest.mark.parametrize("task,T,batch,kw", [("panda_reaching", 200, 3, dict(one_sided_frac=0.25)),
                                             ("panda_reaching", 301, 2, dict(dense_residuals=True, min_N=1)),
                                             ("acrobot", 100, 2, dict(config_id=1, dense_residuals=True, one_sided_frac=0.5)),
                                             ("hopper", 150, 2, dict(dense_residuals=True)),
                                             ("pentabot", 64, 3, dict(one_sided_frac=0.1))])
def test_kp_ordered_payload_gives_the_bytes_of_the_job_lists(task, T, batch, kw, waves):
    """Fused context: key-point ordered payload (RAW backward sweep with one wave per trajectory, the streaming
    differencing kernel otherwise) == job lists + kpilqr_fd_difference, bit for bit; and against the oracle."""
    kw = dict(kw); kw.setdefault("min_N", 5)
    p = synth.make_problem(task=task, T=T, batch=batch, **kw)
    ref = _run(p, True, False, explicit_fd=True)
    assert "fused" in ref["variant"] and np.all(ref["status"] == 0)
    for explicit in (False, True):
        _same(_run(p, True, True, explicit_fd=explicit), ref)
    _same(_run(p, True, False), ref)                              # job lists through kpilqr_iterate (no explicit differencing call)
    for b in range(batch):
        o = pipeline.run_trajectory(p, b)
        assert relerr(ref["K"][b], o["K"]) < 1e-9 and relerr(ref["cost"][b], o["cost_pred"]) < 1e-9


def test_kp_ordered_payload_ragged_lists(waves):
    """Per-DoF lists of very different densities (bisection-shaped), one-sided jobs: every lane walks its own entries."""
    T, dof = 260, 7
    rng = np.random.default_rng(11)
    rows = [synth.bisect_keypoints(rng, dof, T, 1, rng.uniform(0.05, 1.0, dof)) for _ in range(3)]
    p = synth.make_ragged_problem("panda_reaching", T, rows, config_id=4, dense_residuals=True, one_sided_frac=0.3)
    ref = _run(p, True, False, explicit_fd=True)
    got = _run(p, True, True)
    _same(got, ref)
    for b in range(3):
        o = pipeline.run_trajectory(p, b)
        assert relerr(got["K"][b], o["K"]) < 1e-9


@pytest.mark.parametrize("task,T,batch", [("panda_reaching", 120, 2), ("panda_pushing", 90, 2)])
def test_kp_ordered_payload_on_a_context_with_records(task, T, batch):
    """A materialising (non-fused, or tiled) context takes the key-point ordered payload too: A, B, K identical to the job lists."""
    p = synth.make_problem(task=task, T=T, batch=batch, min_N=4, dense_residuals=True, one_sided_frac=0.2)
    a = _run(p, False, False, want_AB=True)
    b = _run(p, False, True, want_AB=True)
    _same(b, a)
    assert np.array_equal(a["A"], b["A"]) and np.array_equal(a["B"], b["B"])


def test_fused_context_materialises_on_demand():
    """A fused context holds no step records; get_AB / interpolate / the cost-derivative hooks allocate and fill them when
    asked, from either payload form, and give the materialising context's bytes."""
    import torch
    p = synth.make_problem(task="panda_reaching", T=150, batch=2, min_N=5, dense_residuals=True, one_sided_frac=0.2)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2) as e:
        synth.upload(e, p)
        e.fd_difference(); e.sync()
        A_kp, B_kp = e.get_AB()
        e.interpolate(); A, B = e.get_AB()
        e.cost_derivs(); lx = e.get_cost_derivs()
    for kp_ordered in (False, True):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
            synth.upload(e, p, kp_ordered=kp_ordered)
            e.iterate(p["lam"], 100, orc.alphas(6))                 # the sweeps run without records
            K0, _ = e.gains()
            A1, B1 = e.get_AB()                                     # key-point columns only
            assert np.array_equal(A1, A_kp) and np.array_equal(B1, B_kp)
            e.interpolate(); A2, B2 = e.get_AB()
            assert np.array_equal(A2, A) and np.array_equal(B2, B)
            e.cost_derivs(); lx2 = e.get_cost_derivs()
            for u, v in zip(lx, lx2):
                assert np.array_equal(u, v)
            e.iterate(p["lam"], 100, orc.alphas(6))                 # and the sweeps are unaffected by the records' presence
            K1, _ = e.gains()
            assert np.array_equal(K0, K1)
    # no records before anyone asks: a fused context of 48 x 3000 Panda steps would hold 645 MB of them
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    q = synth.make_problem(task="panda_reaching", T=3000, batch=1, min_N=5)
    q = synth.tile_problem(q, 48)
    with Engine(q["dof"], q["m"], 3000, q["nr"], batch=48, fused=True) as e:
        synth.upload(e, q, kp_ordered=True)
        e.iterate(q["lam"], 100, orc.alphas(6)); e.sync()
        used = free0 - torch.cuda.mem_get_info()[0]
        assert used < 800e6, used                                    # K, k, residuals + Jacobians (355 MB), payload, kpc: ~735 MB
        e.get_AB()
        assert free0 - torch.cuda.mem_get_info()[0] > used + 600e6   # now the records exist


def test_raw_sweep_pd_failure_and_lambda_retry(waves):
    """The raw sweep may stop at a failed PD check: the retry with a larger lambda differences again and matches the oracle."""
    p = synth.make_problem(task="panda_reaching", T=300, batch=2, min_N=5, dense_residuals=True)
    p["r_u"][:, 120] *= 0.0
    p["r_x"][1, 180:200] *= 40.0
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        synth.upload(e, p, kp_ordered=True)
        lam = np.array([-60.0, 0.1])                                # negative regularisation: not PD at the first check
        st, _ = e.backward(lam, 50)
        o0 = pipeline.run_trajectory(p, 0, lam=-60.0, pd_stride=50, stages=("fd", "interp", "cost", "bwd"))
        assert st[0] == o0["status"] and st[0] > 0 and st[1] == 0
        st, dJ = e.backward(np.array([0.5, 0.1]), 50)               # the reference's retry (iLQR.cpp:435-442)
        assert np.all(st == 0)
        K, k = e.gains()
        cost = e.forward_linear(orc.alphas(6))
    for b, l in enumerate((0.5, 0.1)):
        o = pipeline.run_trajectory(p, b, lam=l, pd_stride=50)
        assert relerr(K[b], o["K"]) < 1e-9 and relerr(cost[b], o["cost_pred"]) < 1e-9


def test_kp_ordered_payload_argument_checks():
    p = synth.make_problem(task="panda_reaching", T=60, batch=2, min_N=5)
    xp, xm, mode = synth.kp_ordered_payload(p)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=2, fused=True) as e:
        s = e.fd_kp_slab(xp, xm, mode)
        with pytest.raises(KpilqrError) as ei:
            e.upload_fd_kp(s)                                       # before the key-points it is ordered by
        assert ei.value.code == -5
        e.set_keypoints_rows(p["kp_rows"])
        bad = dict(s); bad["entries"] = s["entries"] - 1
        with pytest.raises(KpilqrError) as ei:
            e.upload_fd_kp(bad)
        assert ei.value.code == -1
        e.upload_fd_kp(s)
        # new key-points invalidate the payload: the sweeps then have nothing to difference and must say so ... they run on
        # the (zeroed) column store rather than on a payload laid out by other lists
        q = synth.make_problem(task="panda_reaching", T=60, batch=2, min_N=3)
        e.set_keypoints_rows(q["kp_rows"])
        e.upload_residuals(p["r"], p["r_x"], None, p["w_run"], p["w_term"]); e.upload_nominal(p["u_nom"], p["ctrl_lim"])
        e.iterate(0.1, 100, orc.alphas(6))
        K, _ = e.gains()
        assert np.all(np.isfinite(K))


@pytest.mark.parametrize("fused", [True, False])
def test_streamed_iteration_with_the_kp_ordered_payload(fused):
    p = synth.make_problem(task="panda_reaching", T=200, batch=11, min_N=5, dense_residuals=True, one_sided_frac=0.2)
    ref = _run(p, fused, False)
    xp, xm, mode = synth.kp_ordered_payload(p)
    for nchunks in (1, 3, 5):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=11, fused=fused) as e:
            e.set_keypoints_rows(p["kp_rows"])
            e.upload_residuals(None, None, None, p["w_run"], p["w_term"])
            e.upload_nominal(None, p["ctrl_lim"])
            e.forward_linear(orc.alphas(6), fetch=False)
            s = e.fd_kp_slab(xp, xm, mode)
            pin = {}
            for name in ("r", "r_x", "r_u", "u_nom"):
                pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
            lam = e.pinned(11); lam[:] = p["lam"]
            K = e.pinned(ref["K"].shape); k = e.pinned(ref["k"].shape); cp = e.pinned((11, 6)); st = e.pinned(11, np.int32)
            for _ in range(2):
                e.iterate_streamed(fd_kp=s, eps=p["eps"], lam=lam, K=K, k=k, cost_pred=cp, status=st, nchunks=nchunks, **pin)
            e.sync()
            assert np.all(st == 0)
            assert np.array_equal(K, ref["K"]) and np.array_equal(k, ref["k"]) and np.array_equal(cp, ref["cost"])
            e.iterate_streamed(eps=p["eps"], lam=lam, K=K, k=k, cost_pred=cp, status=st, nchunks=nchunks)    # payload resident: reused
            e.sync()
            assert np.array_equal(K, ref["K"]) and np.array_equal(cp, ref["cost"])


def test_backward_stats_histogram():
    """kpilqr_backward_stats: every step of every trajectory is counted once, the gains are those of kpilqr_backward, and a
    small lambda needs more refresh work than a large one."""
    p = synth.make_problem(task="panda_reaching", T=600, batch=3, min_N=5)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=3, fused=True) as e:
        synth.upload(e, p, kp_ordered=True)
        extra = {}
        for lam in (1e-4, 10.0):
            st, dJ = e.backward(lam, 100)
            K0, k0 = e.gains()
            h = e.backward_stats(100)
            K1, k1 = e.gains()
            assert np.all(st == 0) and np.all(h.sum(1) == p["T"]), h
            assert np.all(h[:, 4] >= p["T"] // 100) and np.all(h[:, 5] == 0)      # the checked steps factorise; nothing indefinite
            assert relerr(K1, K0) < 1e-12 and relerr(k1, k0) < 1e-12
            extra[lam] = int((h[:, 1] + 2 * h[:, 2] + 3 * h[:, 3]).sum())
        assert extra[1e-4] >= extra[10.0]


@pytest.mark.parametrize("fused", [True, False])
def test_host_differenced_columns_give_the_bytes_of_the_fd_payload(fused, waves):
    """kpilqr_upload_kp_columns: the key-point columns differenced on the host (IEEE quotients) and uploaded straight into the
    column store == the key-point ordered FD payload differenced on the device, bit for bit -- gains, costs, and the
    materialised A, B -- on fused and materialising contexts, one-sided jobs included; and through the chunk pipeline."""
    def stages(e, p):
        if not fused:
            e.fd_difference(); e.interpolate(); e.cost_derivs()
        st, dJ = e.backward(p["lam"], 100)
        cost = e.forward_linear(orc.alphas(6))
        K, k = e.gains()
        e.interpolate()
        A, B = e.get_AB()
        return dict(K=K, k=k, delta_J=dJ, cost=cost, A=A, B=B)

    for p in (synth.make_problem(task="panda_reaching", T=160, batch=3, min_N=4, one_sided_frac=0.3, dense_residuals=True),
              synth.make_problem(task="acrobot", T=90, batch=2, min_N=1, config_id=1)):
        xp, xm, mode = synth.kp_ordered_payload(p)
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=fused) as e:
            synth.upload(e, p, kp_ordered=True)
            ref = stages(e, p)
        assert np.any(ref["K"] != 0)
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=p["batch"], fused=fused) as e:
            synth.upload(e, p, kp_ordered=True)            # key-points, residuals, nominal ... and a payload that is replaced:
            e.upload_kp_columns(e.kp_columns(xp, xm, mode, eps=p["eps"]))
            _same(stages(e, p), ref, keys=("K", "k", "delta_J", "cost", "A", "B"))
            # the same columns through kpilqr_iterate_streamed, two chunks
            cols = e.kp_columns(xp, xm, mode, eps=p["eps"])
            Kp = e.pinned(ref["K"].shape); kp_ = e.pinned(ref["k"].shape); cp = e.pinned((p["batch"], 6)); jp = e.pinned((p["batch"],))
            lam = e.pinned((p["batch"],)); lam[:] = p["lam"]
            e.iterate_streamed(kp_cols=cols, lam=lam, K=Kp, k=kp_, cost_pred=cp, delta_J=jp, nchunks=2)
            e.sync()
            _same(dict(K=np.array(Kp), k=np.array(kp_), delta_J=np.array(jp), cost=np.array(cp)), ref)


# ---- constant residual Jacobians (round 4): kpilqr_upload_residual_jacobians_const ------------------------------------------
@pytest.mark.parametrize("task,T,batch,kw", [("panda_reaching", 260, 3, dict(min_N=5)),
                                             ("panda_reaching", 131, 2, dict(min_N=1, one_sided_frac=0.2)),
                                             ("acrobot", 100, 2, dict(config_id=1, min_N=5)),
                                             ("hopper", 150, 2, dict(min_N=4)), ("pentabot", 64, 3, dict(min_N=3))])
def test_constant_residual_jacobians_give_the_bytes_of_the_streamed_form(task, T, batch, kw, waves):
    """One r_x [nr][n] uploaded once (Reaching.cpp:43-54: r = [q - q*, qdot] -> selector rows, r_u = 0) == the same matrix given
    at every step, in every wave organisation and for either payload form -- and against the oracle.  The wave forms that read
    the broadcast copy run the streamed kernels: K, k, delta_J, predicted costs bit for bit.  The one-wave sweeps and the helper
    wave of the consumer / helper pair keep the matrix in registers AND (round 5) the constant block r_x' W r_x as a resident
    tile: Lzz = Cxx + 2 e_n (r_x' W r)' is one product instead of four, in another accumulation order -- those legs are held to
    the streamed form at 1e-12 (and to the oracle at 1e-9 like everything else)."""
    p = synth.make_problem(task=task, T=T, batch=batch, **kw)
    assert p["rx_const"] is not None and not np.any(p["r_u"])
    for kp_ordered in (True, False):
        ref = _run(p, True, kp_ordered)
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=batch, fused=True) as e:
            synth.upload(e, p, kp_ordered=kp_ordered, rx_const=True)
            e.iterate(p["lam"], 100, orc.alphas(6))
            res = e.results()
            K, k = e.gains()
            got = dict(K=K, k=k, status=res["status"], delta_J=res["delta_J"], cost=res["cost_pred"])
            lb, lf = e.last_launch("backward"), e.last_launch("forward")
            _same(got, ref, rtol=1e-12 if ":rxc" in lb else 0.0)
            if waves == "one_wave":
                assert ":w1:" in lb and ":rxc" in lb and ":w1:" in lf and ":rxc" in lf, (lb, lf)
                assert (":raw:" in lb) == kp_ordered
            elif waves == "auto":                          # the consumer / helper pair: the helper keeps the matrix in registers
                assert ":pairh:" in lb and lb.endswith(":ru0:rxc") and (":raw:" in lb) == kp_ordered, lb
            else:
                assert ":rxc" not in lb, lb
            # the materialised cost derivatives see the same Jacobians (broadcast copy on demand)
            e.cost_derivs(); lx_c = e.get_cost_derivs()
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=batch, fused=True) as e:
            synth.upload(e, p, kp_ordered=kp_ordered)
            e.fd_difference(); e.cost_derivs(); lx_s = e.get_cost_derivs()
        for u, v in zip(lx_c, lx_s):
            assert np.array_equal(u, v)
    for b in range(batch):
        o = pipeline.run_trajectory(p, b)
        assert relerr(ref["K"][b], o["K"]) < 1e-9 and relerr(ref["cost"][b], o["cost_pred"]) < 1e-9


@pytest.mark.parametrize("task,T", [("acrobot", 100), ("pentabot", 64), ("panda_reaching", 131)])
def test_residual_row_fetched_in_pairs_ignores_what_lies_behind_it(task, T, waves):
    """Round 5: the one-wave sweeps on uniform key-point sets fetch r_t with two 16-byte requests (residual rows relabelled so that
    a lane's registers hold consecutive residuals).  With an odd residual count (acrobot 5, pentabot 3) the last pair of row t
    reaches one element into row t+1 -- under a zero weight: whatever finite number sits there (here 1e300 in the whole row t = T,
    which no stage uses) changes no bit of any result."""
    p = synth.make_problem(task=task, T=T, batch=3, min_N=4, config_id=1 if task == "acrobot" else 2)
    assert p["rx_const"] is not None
    q = dict(p); q["r"] = p["r"].copy(); q["r"][:, T, :] = 1.0e300
    outs = []
    for prob in (p, q):
        with Engine(p["dof"], p["m"], T, p["nr"], batch=3, fused=True) as e:
            synth.upload(e, prob, kp_ordered=True, rx_const=True)
            e.iterate(p["lam"], 100, orc.alphas(6))
            res = e.results(); K, k = e.gains()
            outs.append(dict(K=K, k=k, delta_J=res["delta_J"], cost=res["cost_pred"]))
    _same(outs[1], outs[0])
    for b in range(3):
        o = pipeline.run_trajectory(p, b)
        assert relerr(outs[0]["K"][b], o["K"]) < 1e-9 and relerr(outs[0]["cost"][b], o["cost_pred"]) < 1e-9


def test_constant_residual_jacobians_mode_switches():
    """Per-step Jacobians uploaded afterwards end the constant mode; a dense constant r_u, a materialising context and the
    chunk pipeline all see the broadcast values; kpilqr_device_ptr(R_X) hands out the broadcast copy."""
    import torch
    p = synth.make_problem(task="panda_reaching", T=120, batch=4, min_N=5)
    rng = np.random.default_rng(5)
    rxc = rng.standard_normal((p["nr"], p["n"])) * 0.3
    ruc = rng.standard_normal((p["nr"], p["m"])) * 0.05
    q = dict(p); q["r_x"] = np.broadcast_to(rxc, p["r_x"].shape).copy(); q["r_u"] = np.broadcast_to(ruc, p["r_u"].shape).copy()
    for fused in (True, False):
        ref = _run(q, fused, False)
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=4, fused=fused) as e:
            synth.upload(e, p, kp_ordered=False)                     # selector rows first ...
            e.upload_residual_jacobians_const(rxc, ruc)              # ... then the dense constants
            e.iterate(p["lam"], 100, orc.alphas(6))
            res = e.results(); K, k = e.gains()
            _same(dict(K=K, k=k, delta_J=res["delta_J"], cost=res["cost_pred"]), ref)
            assert ":rxc" not in e.last_launch("backward")           # dense r_u: the streamed instantiations on the broadcast copy
            dev = torch.as_tensor(e.device_array(4, (4, p["T"] + 1, p["nr"], p["n"])), device="cuda").cpu().numpy()
            assert np.array_equal(dev, q["r_x"])
            # per-step Jacobians again: the constant mode is over
            e.upload_residuals(None, p["r_x"], np.zeros_like(p["r_u"]), None, None)
            e.iterate(p["lam"], 100, orc.alphas(6))
            K2, _ = e.gains()
        back = _run(dict(p, r_u=np.zeros_like(p["r_u"])), fused, False)
        assert np.array_equal(K2, back["K"])
    # chunk pipeline, constants resident ("the shim may upload constants once")
    ref = _run(p, True, True)
    xp, xm, mode = synth.kp_ordered_payload(p)
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=4, fused=True) as e:
        e.set_keypoints_rows(p["kp_rows"])
        e.upload_residuals(None, None, None, p["w_run"], p["w_term"])
        e.upload_residual_jacobians_const(p["rx_const"], None)
        e.upload_nominal(None, p["ctrl_lim"])
        e.forward_linear(orc.alphas(6), fetch=False)
        s = e.fd_kp_slab(xp, xm, mode)
        pin = {}
        for name in ("r", "u_nom"):
            pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
        lam = e.pinned(4); lam[:] = p["lam"]
        K = e.pinned(ref["K"].shape); k = e.pinned(ref["k"].shape); cp = e.pinned((4, 6)); st = e.pinned(4, np.int32)
        e.iterate_streamed(fd_kp=s, eps=p["eps"], lam=lam, K=K, k=k, cost_pred=cp, status=st, nchunks=2, **pin)
        e.sync()
        # (the helper wave keeps r_x' W r_x as a resident tile, round 5: l_xx -- and with it V_xx and K -- in the same bits as the
        # streamed form, l_x in another accumulation order: k and the predicted costs to 1e-12)
        assert np.array_equal(K, ref["K"]) and np.max(np.abs(cp - ref["cost"])) <= 1e-12 * np.max(np.abs(ref["cost"]))
        assert np.max(np.abs(k - ref["k"])) <= 1e-12 * np.max(np.abs(ref["k"]))
    # the arrays above are views of pinned allocations: they outlive the engine (Engine.pinned keeps the block alive)
    assert np.array_equal(K, ref["K"]) and int(st.sum()) == 0


def test_negative_residual_weights_on_the_constant_jacobian_sweeps(monkeypatch):
    """The headline's forward sweep scores on rows of r_x scaled by sqrt|w| (round 5): a NEGATIVE weight -- legal, the reference
    just forms w r^2 (ModelTranslator.cpp:314-327, 552-583) -- must keep its sign (the sign joins outside the root), running and
    terminal weights with signs of their own.  One-wave forms forced (the batch would pick the pairs), against the oracle."""
    monkeypatch.setenv("KPILQR_FUSED_WAVES", "1"); monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    p = synth.make_problem(task="panda_reaching", T=150, batch=3, min_N=5, lam=1.0)
    p["w_run"] = p["w_run"].copy(); p["w_term"] = p["w_term"].copy()
    p["w_run"][[2, 9]] *= -0.05                    # small negative weights on one position and one velocity residual ...
    p["w_term"][[4]] *= -0.01                      # ... and another sign pattern at the terminal step
    with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=3, fused=True) as e:
        synth.upload(e, p, kp_ordered=True, rx_const=True)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results(); K, k = e.gains()
        assert e.last_launch("forward").endswith(":w1:uni:ru0:rxc"), e.last_launch("forward")
    for b in range(3):
        o = pipeline.run_trajectory(p, b)
        assert res["status"][b] == 0 and o["status"] == 0
        assert relerr(K[b], o["K"]) < 1e-9 and relerr(res["cost_pred"][b], o["cost_pred"]) < 1e-9, (b, relerr(res["cost_pred"][b], o["cost_pred"]))


def test_rejected_streamed_call_leaves_the_constant_jacobian_mode_alone():
    """Round-4 advisor: kpilqr_iterate_streamed used to leave the constant-Jacobian mode BEFORE the checks that can still reject
    the call; after a call refused for an unpinned r_x the next sweep read an r_x buffer that never received the broadcast copy.
    Now a rejected call changes nothing: the iteration behind it gives the constant mode's bytes."""
    from trajoptkp_amd.engine import KpilqrError
    p = synth.make_problem(task="panda_reaching", T=120, batch=4, min_N=5)
    xp, xm, mode = synth.kp_ordered_payload(p)
    for fused in (True, False):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=4, fused=fused) as e:
            synth.upload(e, p, kp_ordered=fused, rx_const=True)
            if not fused:
                e.fd_difference(); e.interpolate(); e.cost_derivs()
            e.iterate(p["lam"], 100, orc.alphas(6))
            K0, k0 = e.gains(); c0 = e.results()["cost_pred"].copy()
            lb0 = e.last_launch("backward")
            bad_rx = np.array(p["r_x"], copy=True)                     # ordinary (unpinned) host memory: ERR_ARG
            lam = e.pinned(4); lam[:] = p["lam"]
            with pytest.raises(KpilqrError):
                e.iterate_streamed(r_x=bad_rx, lam=lam, nchunks=2)
            if not fused:
                e.cost_derivs()
            e.iterate(p["lam"], 100, orc.alphas(6))
            K1, k1 = e.gains(); c1 = e.results()["cost_pred"]
            assert e.last_launch("backward") == lb0
            assert np.array_equal(K1, K0) and np.array_equal(k1, k0) and np.array_equal(c1, c0)


# ---- slope store (round 4): per-DoF lists walked on precomputed segment slopes ---------------------------------------------
@pytest.mark.parametrize("payload", ["kp_ordered", "job_lists", "columns"])
def test_per_dof_lists_walk_the_slope_store(payload, monkeypatch):
    """One wave per trajectory, per-DoF (ragged) key-point lists: the general forms of both sweeps take segment start AND slope
    from memory (k_kp_slopes: (next column - column) / gap, the IEEE quotient of KeyPointGenerator.cpp:898-905) -- reported as
    ":ragged:...:slopes" -- and give, for every payload form, the gains of the oracle and bit for bit those of the wave pair /
    triple, which still divide at the crossing.  Also through the chunk pipeline (every chunk makes the slopes of its own entries)."""
    T, dof, B = 280, 7, 5
    rng = np.random.default_rng(23)
    dens = rng.uniform(0.03, 1.0, dof); dens[2] = 0.0; dens[5] = 1.0       # one DoF only at 0 and T-1, one at every step
    rows = [synth.bisect_keypoints(rng, dof, T, 1, np.roll(dens, b)) for b in range(B)]
    p = synth.make_ragged_problem("panda_reaching", T, rows, config_id=4, dense_residuals=True, one_sided_frac=0.25)
    xp, xm, mode = synth.kp_ordered_payload(p)

    def run(env):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
            synth.upload(e, p, kp_ordered=payload != "job_lists")
            if payload == "columns":
                e.upload_kp_columns(e.kp_columns(xp, xm, mode, eps=p["eps"]))
            e.iterate(p["lam"], 100, orc.alphas(6))
            res = e.results(); K, k = e.gains()
            out = dict(K=K, k=k, delta_J=res["delta_J"], cost=res["cost_pred"], lb=e.last_launch("backward"), lf=e.last_launch("forward"))
        for k_ in env:
            monkeypatch.delenv(k_)
        return out

    one = run({"KPILQR_FUSED_WAVES": "1", "KPILQR_FUSED_FWD_WAVES": "1"})
    assert ":w1:kpc:ragged" in one["lb"] and one["lb"].endswith(":slopes") and one["lf"].endswith(":slopes"), (one["lb"], one["lf"])
    trip = run({})                                              # B = 5: the consumer / helper pair, its helper on the slope store too
    assert ":pairh:" in trip["lb"] and trip["lb"].endswith(":slopes"), trip["lb"]
    assert ":triple:ragged" in trip["lf"], trip["lf"]          # forward: the uniform pair left at once, the triple behind it ran
    # (the helper on the slope store and the one-wave general form on it: the same slopes -- the correctly rounded quotients of
    # KeyPointGenerator.cpp:898-905 -- in other wave organisations)
    assert all(np.max(np.abs(trip[key] - one[key])) <= 1e-11 * np.max(np.abs(one[key])) for key in ("K", "k", "delta_J"))
    for b in range(B):
        o = pipeline.run_trajectory(p, b)
        assert relerr(one["K"][b], o["K"]) < 1e-9 and relerr(one["cost"][b], o["cost_pred"]) < 1e-9
        assert relerr(trip["K"][b], one["K"][b]) < 1e-11
    if payload == "kp_ordered":
        # the chunk pipeline: every chunk differences and makes the slopes of its own entry range
        monkeypatch.setenv("KPILQR_FUSED_WAVES", "1"); monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
        with Engine(p["dof"], p["m"], T, p["nr"], batch=B, fused=True) as e:
            e.set_keypoints_rows(p["kp_rows"])
            e.upload_residuals(None, None, None, p["w_run"], p["w_term"]); e.upload_nominal(None, p["ctrl_lim"])
            e.forward_linear(orc.alphas(6), fetch=False)
            s = e.fd_kp_slab(xp, xm, mode)
            pin = {}
            for name in ("r", "r_x", "r_u", "u_nom"):
                pin[name] = e.pinned(p[name].shape); pin[name][...] = p[name]
            lam = e.pinned(B); lam[:] = p["lam"]
            K = e.pinned(one["K"].shape); cp = e.pinned((B, 6))
            for _ in range(2):
                e.iterate_streamed(fd_kp=s, eps=p["eps"], lam=lam, K=K, cost_pred=cp, nchunks=3, **pin)
            e.sync()
            assert np.array_equal(K, one["K"]) and np.array_equal(cp, one["cost"])


def test_general_forms_forced_on_uniform_lists(monkeypatch):
    """KPILQR_FUSED_UNI=0 (diagnostic): every key-point set counts as per-DoF lists, so the GENERAL forms of the one-wave sweeps --
    the raw backward sweep that divides at its crossings, the forward sweep on the slope store -- run on set_interval lists too;
    same gains as the segment-loop forms (1e-12) and the oracle's (1e-9), for either payload."""
    p = synth.make_problem(task="panda_reaching", T=230, batch=3, min_N=4, one_sided_frac=0.2)
    monkeypatch.setenv("KPILQR_FUSED_WAVES", "1"); monkeypatch.setenv("KPILQR_FUSED_FWD_WAVES", "1")
    ref = _run(p, True, True)
    monkeypatch.setenv("KPILQR_FUSED_UNI", "0")
    for kp_ordered in (True, False):
        with Engine(p["dof"], p["m"], p["T"], p["nr"], batch=3, fused=True) as e:
            synth.upload(e, p, kp_ordered=kp_ordered)
            e.iterate(p["lam"], 100, orc.alphas(6))
            res = e.results(); K, k = e.gains()
            lb, lf = e.last_launch("backward"), e.last_launch("forward")
        assert ":ragged" in lb and ":ragged" in lf and ((":raw:" in lb) == kp_ordered), (lb, lf)
        assert relerr(K, ref["K"]) < 1e-12 and relerr(res["cost_pred"], ref["cost"]) < 1e-11
    for b in range(3):
        o = pipeline.run_trajectory(p, b)
        assert relerr(ref["K"][b], o["K"]) < 1e-9


@pytest.mark.parametrize("ragged_pair", ["0", "1"])
def test_per_dof_lists_between_256_and_512_trajectories(monkeypatch, ragged_pair):
    """256 < B <= 512 with per-DoF key-point lists: backward = the consumer / helper pair on the slope store, forward = one wave
    per trajectory (general form) or, with KPILQR_FWD_RAGGED_PAIR=1, the state / cost+staging wave pair behind the uniform pair
    (which leaves at once): the forms are asserted, a few trajectories against the oracle, the two forward forms against each other."""
    monkeypatch.setenv("KPILQR_FWD_RAGGED_PAIR", ragged_pair)
    T, dof = 90, 7
    rng = np.random.default_rng(5)
    rows = [synth.bisect_keypoints(rng, dof, T, 2, rng.uniform(0.0, 1.0, dof)) for _ in range(8)]
    p0 = synth.make_ragged_problem("panda_reaching", T, rows, config_id=6, dense_residuals=False)
    p = synth.tile_problem(p0, 33)                          # B = 264
    with Engine(p["dof"], p["m"], T, p["nr"], batch=p["batch"], fused=True) as e:
        synth.upload(e, p, kp_ordered=True)
        e.iterate(p["lam"], 100, orc.alphas(6))
        res = e.results(); K, k = e.gains()
        lb, lf = e.last_launch("backward"), e.last_launch("forward")
    assert ":pairh:kpc:ragged" in lb and lb.endswith(":slopes"), lb
    assert (":pair:ragged" in lf) if ragged_pair == "1" else (":w1:ragged" in lf), lf
    for b in (0, 5, 8 * 32 + 3):
        o = pipeline.run_trajectory(p0, b % 8)
        assert res["status"][b] == 0 and relerr(K[b], o["K"]) < 1e-9 and relerr(res["cost_pred"][b], o["cost_pred"]) < 1e-9
    assert np.array_equal(K[3], K[8 * 32 + 3]) and np.array_equal(res["cost_pred"][3], res["cost_pred"][8 * 32 + 3])
