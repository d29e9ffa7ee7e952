#!/usr/bin/env python3
"""bench.py -- iLQR iterations/sec on the keypoint-iLQR hot path (BASELINE.json metric).

One "step" = one whole GPU-side iLQR iteration for every trajectory of the batch:
    FD differencing (a2) -> interpolation (a4) -> cost derivatives (a6) -> backward pass (a7, one pass at a valid lambda)
    -> linearised forward pass over the 6 line-search alphas (a8)
By default (KPILQR_FLAG_FUSED) that is TWO launches: the backward sweep differences the key-point ordered FD payload at its
segment crossings and evaluates a4 and a6 inside, the forward sweep reads the columns it leaves behind (A, B, l_* never
written to HBM, no step records); `--unfused` times the materialising five-kernel pipeline.  All inputs (host FD results,
residuals and their Jacobians, nominal controls) are resident in HBM when the timed region starts.

Workload: BASELINE configs[3] -- Franka Panda 7-DoF reaching, T=3000, set-interval key-points every 5 steps, a GLOBAL
batch of 1024 independent trajectories (MPC replans) with 1024 DISTINCT seeds, sharded in contiguous blocks over the N ranks
(`--global-batch`, strong scaling: N=1 runs all 1024 on one GPU).  `--weak` keeps `--batch` trajectories PER GPU instead.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE compact JSON line LAST on stdout (<= 4 KB: compact_line()) and writes everything it measured to a detail
file (bench_detail.json beside this script, or --detail PATH).  Besides the contract's keys the results carry
  roofline        dominant kernel (backward sweep): the binding resource is the SIMD's FP64 matrix pipe, so `bound` is
                  "mfma" (algorithmic a7 flops / HIP-event launch time vs the FP64 MFMA peak); `hbm` inside it gives the
                  HBM fractions (the kernel's own compulsory bytes, the PMC traffic, and SURVEY 8(d)'s riccati figure)
  parity_check    K, k, predicted costs of the first seeds against the CPU oracle (outside the timed region)
  lambda_sweep    both sweeps over the reference's regularisation range and on a mixed batch, with the refresh histogram
  pcie_inclusive  SURVEY 8(d): the same iteration with the FD payload / residuals re-uploaded and K,k downloaded every
                  iteration (kpilqr_iterate_streamed), full payload and resident-Jacobian form           (N=1 only)
  secondary_configs  BASELINE configs[1], [2], [4] (and the humanoid shape) with their own roofline objects (N=1 only)
  strong_scaling_projection  configs[3] shards (512 / 256 / 128 trajectories) timed on this one GPU, x N     (N=1 only)
  materialising_pipeline, cpu_baseline (the CPU oracle = line-faithful port, timed on this box's host cores; N=1 only)
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The chunk pipeline of kpilqr_iterate_streamed (pcie_inclusive) wants a hardware queue per stream: context stream + 3 chunk
# streams + this script's torch stream exceed the runtime's default of 4.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# FP64 matrix pipe: v_mfma_f64_16x16x4_f64 = 2048 flop in 64 busy cycles = 32 flop/clk/SIMD (measured,
# profiles/r01_fp64_mfma_probe.txt) x 1024 SIMDs x 2.4 GHz; the guide has no FP64 row, this is the vector FP64 peak too
FP64_PEAK_TFLOPS = 1024 * 32 * 2.4e9 / 1e12
# Residual Jacobians of the ragged-key-point side workloads (configs[2], configs[4]): True = drawn independently at every step (the
# default, and a worst case: the running inverse of Q_uu never applies, every step factorises), "smooth" = functions of the state
# that move slowly along the trajectory and none of the control, the structure of the reference's tasks (synth.smooth_residual_jacobians)
RESIDUAL_MODEL = "smooth" if os.environ.get("KPILQR_BENCH_RESIDUALS") == "smooth" else True
RESIDUAL_DESC = {True: "drawn independently at every step (worst case: every step factorises)",
                 "smooth": "smooth in time, none of the control (the structure of the reference's tasks, TwoDPushing.cpp:291-352)"}


def algorithmic_bytes(dof, m, nr, T, Kp, n_alpha):
    """compulsory FP64 stage I/O per trajectory-iteration (A, B, l_* materialised once)."""
    n = 2 * dof
    return dict(
        fd_difference=8 * Kp * ((2 * dof + m) * 2 * n + (n * n + n * m)),
        interpolate=8 * (Kp * (n * n + n * m) + T * (n * n + n * m)),
        cost_derivs=8 * ((T + 1) * nr * (1 + n + m) + T * (n + n * n + m + m * m)),
        backward=8 * T * ((n * n + n * m + n + n * n + m + m * m) + (m * n + m)),
        forward=8 * T * (n * n + n * m + m * n + m + n + n * n + m + m * m) + 8 * n_alpha,
    )


def fused_bytes(n, m, nr, T, pairs, n_alpha):
    """What the fused sweeps must move: key-point columns + residuals/Jacobians in, gains out (backward); the same plus
    gains and nominal controls in (forward).  pairs = (key-point (time, DoF) pairs, those with DoF < num_ctrl) of the
    trajectory: two A columns per pair, one B column per actuated pair."""
    src = 8 * (pairs[0] * 2 * n + pairs[1] * n + T * nr * (1 + n + m))
    return dict(backward=src + 8 * T * (m * n + m), forward=src + 8 * T * (m * n + m + m) + 8 * n_alpha)


def flops_a7(n, m):
    """Algorithmic flops of one backward step (iLQR.cpp:567-613; SURVEY 8(d): 30.3 k Panda, 67.5 k n=20, 1.24 M n=62)."""
    return 4 * n ** 3 + 10 * m * n * n + 6 * m * m * n + 2 * n * n + 2 * m * n + 2 * m ** 3 + 2 * m * m


_ORC_NATIVE = False


def _oracle_native():
    """The CPU oracle built -O3 -march=native for THIS host (once per process), loaded as oracle.oracle's library."""
    global _ORC_NATIVE
    import tempfile
    from oracle import oracle as orc
    if not _ORC_NATIVE:
        path = orc.build(native=True, out_dir=tempfile.mkdtemp(prefix="kpilqr_oracle_"))
        orc._LIB = orc.lib(path)
        _ORC_NATIVE = True
    return orc


def host_cores():
    """(online, in this process's affinity mask, cgroup CPU quota in cores or None, the share a thread pool should use)."""
    online = os.cpu_count() or 1
    avail = online
    try:
        avail = len(os.sched_getaffinity(0))             # the cores this process may run on (a 1-GPU box's share of the host)
    except Exception:
        pass
    # ... and the CPU TIME it is given: a 1-GPU box is a container on a 256-core host whose affinity mask shows every core while its
    # cgroup quota is a 16-core share (round 4, measured: 256 threads 364 it/s, 16 threads 721 it/s on such a box)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                        # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    share = avail if quota is None else max(1, min(avail, int(round(quota))))
    return online, avail, quota, share


def cpu_baseline(task, T, min_N, reps_per_thread=60):
    """Times the CPU oracle (oracle/kpilqr_oracle.c = line-faithful port of the reference, built here with
    -O3 -march=native) on this host: whole trajectory-iterations (the same five stages) run by a pthread
    pool inside the C library, one independent trajectory per thread at a time -- the batch analogue of
    the reference's hardware_concurrency() thread pools.  Thread count = every core this process may run on (printed)."""
    from trajoptkp_amd import synth
    orc = _oracle_native()
    online, avail, quota, share = host_cores()
    p = synth.make_problem(task=task, T=T, batch=1, min_N=min_N)
    orc.iteration_batch_seconds(p, 0, 1, 1)                                # warm-up
    t_single = orc.iteration_batch_seconds(p, 0, 1, 5) / 5
    # Thread counts tried: KPILQR_CPU_THREADS if given; else every core of the share (as the reference's pools do,
    # hardware_concurrency() - 1: Optimiser.cpp:227,280) and the 16 threads rounds 1-3 quoted -- `value` is the BEST of them,
    # `cores` the thread count that gave it.  ~1 s of wall clock per run (the whole sample is bounded at ~20 s of CPU per thread count).
    forced = os.environ.get("KPILQR_CPU_THREADS")
    counts = [max(1, int(forced))] if forced else sorted({share, min(16, avail)})
    tried = {}
    for c_ in counts:
        reps = reps_per_thread if c_ <= 32 else max(8, reps_per_thread * 32 // c_)     # (bounded even if the threads outnumber the CPU time given)
        wall = orc.iteration_batch_seconds(p, 0, c_, reps)
        tried[c_] = (c_ * reps / wall, c_ * reps, wall)
    cores = max(tried, key=lambda c_: tried[c_][0])
    val, n_traj, wall = tried[cores]
    return {"value": val, "unit": "trajectory-iterations/s", "cores": cores, "cores_available": avail, "cores_online": online,
            "cgroup_cpu_quota_cores": quota, "kind": "port",
            "threads_tried": {str(c_): v[0] for c_, v in tried.items()},
            "sample": f"{n_traj} trajectory-iterations ({task}, T={T}, key-points every {min_N}) on {cores} pthreads "
                      f"({wall:.1f} s wall; host cores online {online}, in this process's affinity mask {avail}, cgroup CPU quota "
                      f"{'none' if quota is None else f'{quota:g} cores'}); single thread: {1.0 / t_single:.2f} it/s",
            "sample_short": f"{n_traj} traj-iterations ({task} T={T}) on {cores} pthreads, {wall:.1f} s; online {online}, quota {quota}",
            "single_thread_value": 1.0 / t_single}


def cpu_baseline_problem(p0, batch, budget_s=8.0):
    """The CPU oracle on a side workload (BASELINE configs[1], [2], [4]): trajectory 0 of p0 (the first of the tiled seeds), run by
    min(batch, the host's CPU share) pthreads -- a batch of `batch` independent trajectories cannot use more threads than it has
    trajectories (configs[1]: ONE).  Bounded: one single-thread repetition as warm-up and estimate, then repetitions for ~budget_s."""
    orc = _oracle_native()
    online, avail, quota, share = host_cores()
    threads = max(1, min(int(batch), share))
    t1 = orc.iteration_batch_seconds(p0, 0, 1, 1)
    reps = int(max(1, min(60, budget_s / max(t1, 1e-6))))
    wall = orc.iteration_batch_seconds(p0, 0, threads, reps)
    return {"value": threads * reps / wall, "unit": "trajectory-iterations/s", "cores": threads, "cgroup_cpu_quota_cores": quota, "kind": "port",
            "single_thread_value": 1.0 / t1,
            "sample": f"{threads * reps} trajectory-iterations of the workload's first seed on {threads} pthreads ({wall:.1f} s wall)"}


# ---- the bench line -----------------------------------------------------------------------------------------------------
# The LAST stdout line of a run is ONE compact JSON object (target <= 4 KB, every string <= 120 characters: the driver's
# record keeps scalars of `config` / `roofline` / `cpu_baseline` and cuts strings there).  Everything else this script
# measures -- secondary configs, the regularisation sweep, the materialising pipeline, the PCIe tables, CPU baselines per
# BASELINE config -- goes to a DETAIL FILE (bench_detail.json beside this script, or --detail PATH; its path is in the line).
LINE_LIMIT = 4096


def _sig(x, digits=6):
    """floats to `digits` significant digits, recursively (the line is for reading; the detail file keeps full precision)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return x if (x != x or x in (float("inf"), float("-inf")) or x == 0.0) else float(f"{x:.{digits}g}")
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _cut(s_, n=120):
    return s_ if not isinstance(s_, str) or len(s_) <= n else s_[:n - 1] + "~"


def compact_line(out, detail_path=None):
    """The bench line of a full result dict `out` (what main() assembles): the contract's keys, `roofline`, `cpu_baseline`,
    the PCIe-inclusive rate, the per-step-Jacobian rate and the scaling projections as first-class scalars.  Pure function
    (tests/test_bench_line.py builds it from a canned dict on the CPU)."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    cfg, roof, cpu = out.get("config", {}), out.get("roofline") or {}, out.get("cpu_baseline")
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _cut(cfg.get("workload_short") or cfg.get("workload", "")),
                      "batch_per_gpu": cfg.get("batch_per_gpu"), "global_batch": cfg.get("global_batch"), "horizon": cfg.get("horizon"),
                      "launched": g(cfg, "launched"),
                      "launched_backward": _cut(g(cfg, "launched", "backward")), "launched_forward": _cut(g(cfg, "launched", "forward")),
                      "residual_jacobians": cfg.get("residual_jacobians"), "residual_jacobians_source": _cut(cfg.get("residual_jacobians_source")),
                      "valid_backward_passes_rank0": cfg.get("valid_backward_passes_rank0"), "parallelism": cfg.get("parallelism")}
    line["rccl_ranks"] = out.get("rccl_ranks")
    line["collective"] = _cut(out.get("collective"))
    line["stage_ms"] = out.get("stage_ms")
    hbm = roof.get("hbm") or {}
    line["roofline"] = {"bound": roof.get("bound"), "kernel": _cut(roof.get("kernel")), "achieved": roof.get("achieved"), "peak": roof.get("peak"),
                        "unit": roof.get("unit"), "frac": roof.get("frac"), "traffic": roof.get("traffic"),
                        "traffic_source": _cut(roof.get("traffic_source_short") or roof.get("traffic_source")),
                        "avg_launch_ms": roof.get("avg_launch_ms"), "flops_per_trajectory_step": roof.get("flops_per_trajectory_step"),
                        "issued_mfma_per_trajectory_step": g(roof, "issued_mfma", "mfma_16x16x4_per_trajectory_step"),
                        "kernel_compulsory_bytes_per_launch": hbm.get("kernel_compulsory_bytes_per_launch"),
                        "frac_of_hbm_peak": hbm.get("frac_of_hbm_peak"), "traffic_frac_of_hbm_peak": hbm.get("traffic_frac_of_hbm_peak"),
                        "hbm": {k: hbm.get(k) for k in ("kernel_compulsory_bytes_per_launch", "frac_of_hbm_peak", "traffic_frac_of_hbm_peak")}}
    if isinstance(cpu, dict) and "error" not in cpu:
        line["cpu_baseline"] = {k: (_cut(cpu.get(k)) if k == "sample" else cpu.get(k)) for k in
                                ("value", "unit", "cores", "cores_available", "cgroup_cpu_quota_cores", "kind", "single_thread_value", "sample")}
        line["cpu_baseline"]["sample"] = _cut(cpu.get("sample_short") or cpu.get("sample"))
    elif cpu is not None:
        line["cpu_baseline"] = {"error": _cut(str(cpu.get("error")))}
    for k in ("gpu_over_cpu", "gpu_over_cpu_single_thread"):
        if k in out: line[k] = out[k]
    pc = out.get("parity_check") or {}
    line["parity_check"] = {k: pc.get(k) for k in ("trajectories_checked", "max_rel_err_K", "max_rel_err_cost_pred", "pass")}
    # the same iteration with the payload crossing the link every iteration -- what a drop-in caller gets
    px = out.get("pcie_inclusive_b1024") if isinstance(g(out, "pcie_inclusive_b1024", "full_payload_constant_jacobians"), dict) else out.get("pcie_inclusive")
    if isinstance(g(px, "full_payload_constant_jacobians"), dict):
        a = px["full_payload_constant_jacobians"]
        line["value_pcie_inclusive"] = a.get("value")
        line["pcie_inclusive"] = {"batch": px.get("batch"), "payload": "x+,x- key-point records + residuals up, K,k down; constant r_x resident",
                                  "ms_per_iteration": a.get("ms_per_iteration"), "link_GBps": a.get("link_GBps"),
                                  "host_differenced_columns_value": g(px, "full_payload_constant_jacobians_host_differenced_columns", "value"),
                                  "per_step_jacobians_value": g(px, "full_payload", "value")}
    if isinstance(out.get("per_step_residual_jacobians"), dict) and "value" in out["per_step_residual_jacobians"]:
        ps = out["per_step_residual_jacobians"]
        line["value_per_step_jacobians"] = ps["value"]
        line["per_step_jacobians"] = {"ms_per_step": ps.get("ms_per_step"), "stage_ms": ps.get("stage_ms"), "roofline_frac": g(ps, "roofline", "frac")}
    ssp = g(out, "strong_scaling_projection", "n_gpus")
    if isinstance(ssp, dict):
        line["strong_scaling_projection"] = {k: g(v, "projected_value") for k, v in ssp.items()}
        line["strong_scaling_projection"]["note"] = "global batch 1024 / N, shards timed on THIS GPU x N: projection, not a measurement"
    wsp = g(out, "weak_scaling_projection", "n_gpus")
    if isinstance(wsp, dict):
        line["weak_scaling_projection"] = {k: g(v, "projected_value") for k, v in wsp.items()}
    if isinstance(out.get("weak_scaling"), dict):
        line["weak_scaling"] = {k: out["weak_scaling"].get(k) for k in ("batch_per_gpu", "global_batch", "value", "ms_per_step")}
    sec = out.get("secondary_configs")
    if isinstance(sec, dict):          # one number per BASELINE config: [value, gpu_over_cpu]; the rest is in the detail file
        line["configs"] = {re.sub(r"\s.*", "", k) if k.startswith("configs[") and "smooth" not in k else None: [g(v, "value"), g(v, "gpu_over_cpu")]
                           for k, v in sec.items()}
        line["configs"].pop(None, None)
    if detail_path:
        line["detail"] = detail_path
    line = _sig(line)
    # never longer than the limit: drop the optional blocks, last first
    for k in ("configs", "weak_scaling_projection", "per_step_jacobians", "pcie_inclusive", "gpu_over_cpu_single_thread"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def emit(out, detail_path, rank0_stdout=None):
    """Writes the detail file, a readable digest of the side measurements to stderr, and the compact line LAST on stdout."""
    rel = None
    try:
        with open(detail_path, "w") as f:
            json.dump(out, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else repr(o))
        rel = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    except Exception as ex:
        print(f"bench.py: detail file not written: {ex!r}", file=sys.stderr)
    line = compact_line(out, rel)
    try:
        sec = out.get("secondary_configs") or {}
        for k, v in sec.items():
            if isinstance(v, dict) and "value" in v:
                print(f"[bench] {k}: {v['value']:.1f} traj-it/s, {v['ms_per_step']:.2f} ms {v.get('stage_ms')} cpu x{v.get('gpu_over_cpu')}", file=sys.stderr)
    except Exception:
        pass
    sys.stderr.flush()
    print(json.dumps(line), file=rank0_stdout or sys.stdout, flush=True)
    return line


# ---- problems ---------------------------------------------------------------------------------------------------------
_BIG = ("xplus", "xminus", "xnom", "r", "r_x", "r_u", "u_nom")


def _shared(shape, dtype=np.float64):
    """numpy array in anonymous shared memory: forked workers fill their slice in place (nothing is pickled back)."""
    import mmap
    nbytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 1)
    return np.frombuffer(mmap.mmap(-1, nbytes), dtype=dtype, count=int(np.prod(shape))).reshape(shape)


_FILL = None


def _fill_entry(lo):
    return _FILL(lo)


def distinct_problem(task, T, B, min_N, first_b=0, workers=None, cache=None):
    """B trajectories with B DISTINCT seeds (synth.seed_for(2, first_b + b)): what a batch of MPC replans / initial
    conditions looks like.  Generated by forked workers, a few seeds each, straight into shared arrays; must run BEFORE
    anything initialises HIP in this process (fork).  cache: a directory -- the big arrays are kept there as .npy files and
    memory-mapped by later runs (the rocprofv3 passes of tools/collect_profiles.sh, whose preloaded profiler has initialised
    the GPU before this script starts, must not fork)."""
    import multiprocessing as mp
    from trajoptkp_amd import synth
    global _FILL
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0)) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    if B <= 2:
        return synth.make_problem(task=task, T=T, batch=B, min_N=min_N, first_b=first_b)
    # the generator's own text in the key: a change to synth.py (seeds, residual model, dynamics) must not be served stale arrays
    import hashlib
    gen = hashlib.sha256(open(synth.__file__.replace(".pyc", ".py"), "rb").read()).hexdigest()[:10]
    stem = os.path.join(cache, f"{task}_T{T}_B{B}_N{min_N}_b{first_b}_g{gen}_") if cache else None
    if stem and all(os.path.exists(stem + k + ".npy") for k in _BIG):
        out = {k: np.load(stem + k + ".npy", mmap_mode="r") for k in _BIG}
        return _assemble(synth.make_problem(task=task, T=T, batch=1, min_N=min_N, first_b=first_b), out, B)
    probe = synth.make_problem(task=task, T=T, batch=1, min_N=min_N, first_b=first_b)
    nj, nn = len(probe["job_b"]), len(probe["xnom"])
    out = {k: _shared((B * probe[k].shape[0],) + probe[k].shape[1:]) for k in _BIG}
    chunk = max(1, min(16, (B + 2 * workers - 1) // (2 * workers)))

    def fill(lo):
        hi = min(lo + chunk, B)
        q = synth.make_problem(task=task, T=T, batch=hi - lo, min_N=min_N, first_b=first_b + lo)
        for k in _BIG:
            per = probe[k].shape[0]
            out[k][lo * per:hi * per] = q[k]
        return lo

    if workers == 1 or os.environ.get("KPILQR_BENCH_NOFORK"):
        for lo in range(0, B, chunk):
            fill(lo)
    else:
        _FILL = fill
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_fill_entry, list(range(0, B, chunk)))
        _FILL = None
    if stem:
        os.makedirs(cache, exist_ok=True)
        for k in _BIG:
            np.save(stem + k + ".npy.tmp.npy", out[k]); os.replace(stem + k + ".npy.tmp.npy", stem + k + ".npy")
    return _assemble(probe, out, B)


def _assemble(probe, out, B):
    nj, nn = len(probe["job_b"]), len(probe["xnom"])
    p = {k: v for k, v in probe.items() if k not in ("A_kp", "B_kp")}
    p["batch"] = B
    for k in _BIG:
        p[k] = out[k]
    p["job_b"] = np.repeat(np.arange(B, dtype=np.int32), nj)
    for k in ("job_t", "job_col", "job_mode"):
        p[k] = np.tile(probe[k], B)
    p["job_nom"] = (np.tile(probe["job_nom"], B) + np.repeat(np.arange(B, dtype=np.int32) * nn, nj)).astype(np.int32)
    p["kp_rows"] = probe["kp_rows"] * B
    return p


def slice_problem(p, B):
    """The first B trajectories of a problem whose jobs are ordered by trajectory (shards of the headline batch)."""
    B0 = p["batch"]
    if B == B0:
        return p
    nj, nn = len(p["job_b"]) // B0, len(p["xnom"]) // B0
    q = dict(p)
    q["batch"] = B
    for k in ("job_b", "job_t", "job_col", "job_mode", "job_nom", "xplus", "xminus"):
        q[k] = p[k][:B * nj]
    q["xnom"] = p["xnom"][:B * nn]
    for k in ("r", "r_x", "r_u", "u_nom"):
        q[k] = p[k][:B]
    q["kp_rows"] = p["kp_rows"][:B]
    return q


def build_problem(kind, B, T, min_N, task, first_b=0, distinct=True, cache=None, residuals=None):
    """Returns (problem of B trajectories, the problem of its first few trajectories for the oracle check, description)."""
    from trajoptkp_amd import synth
    if kind == "set_interval":
        if distinct:
            p = distinct_problem(task, T, B, min_N, first_b, cache=cache)
            p0 = synth.make_problem(task=task, T=T, batch=min(8, B), min_N=min_N, first_b=first_b)
            return p, p0, f"{task} T={T} set_interval({min_N}), {B} distinct seeds"
        uniq = min(8, B)
        if B % uniq:
            uniq = 1
        p0 = synth.make_problem(task=task, T=T, batch=uniq, min_N=min_N, first_b=first_b)
        desc = f"{task} T={T} set_interval({min_N}), {uniq} distinct seeds tiled"
    elif kind == "adaptive_jerk":          # BASELINE configs[2]: contact trajectory, jerk thresholds 10 (joints) / 1 (body)
        from trajoptkp_amd import host        # the product's own KeypointGenerator (host C++) places the key-points
        uniq = min(8, B)
        dof, dt = synth.TASKS[task]["dof"], synth.TASKS[task]["dt"]
        thr = np.array([10.0] * min(7, dof) + [1.0] * max(0, dof - 7))
        rows = [host.keypoints("adaptive_jerk", dof, T, 1, 100, thresholds=thr, dt=dt,
                               X=synth.contact_trajectory(np.random.default_rng(synth.seed_for(3, b) + 17), dof, T, dt))[:2]
                for b in range(uniq)]
        rm = residuals if residuals is not None else RESIDUAL_MODEL
        p0 = synth.make_ragged_problem(task, T, rows, config_id=3, dense_residuals=rm)
        desc = f"{task} T={T} adaptive_jerk(min_N=1,max_N=100) ragged key-points, {uniq} distinct seeds tiled, residual Jacobians {RESIDUAL_DESC[rm]}"
    elif kind in ("reach_velocity_change", "reach_adaptive_jerk"):
        # The reference's OWN default for reaching: keypointMethod "velocity_change", minN 1, maxN 50, magVelThresholds
        # [2, 2, 2, 2, 0.5, 0.5, 0.5] (TaskConfigs/free_motion/reaching.yaml:6-8,18; KeyPointGenerator.cpp:642-728), and
        # adaptive_jerk with the same interval bounds and jointJerkThresholds 10 (:17; :341-382,730-770) -- per-DoF (ragged)
        # lists placed by the product's host KeypointGenerator on a synthetic reach (cubic spline + a small tracking wiggle).
        from trajoptkp_amd import host
        uniq = min(8, B)
        dof, dt = synth.TASKS[task]["dof"], synth.TASKS[task]["dt"]
        method = kind[len("reach_"):]
        thr = np.array([2.0, 2.0, 2.0, 2.0, 0.5, 0.5, 0.5][:dof]) if method == "velocity_change" else np.full(dof, 10.0)
        rows = [host.keypoints(method, dof, T, 1, 50, thresholds=thr, dt=dt,
                               X=synth.contact_trajectory(np.random.default_rng(synth.seed_for(6, b) + 5), dof, T, dt))[:2]
                for b in range(uniq)]
        p0 = synth.make_ragged_problem(task, T, rows, config_id=6, dense_residuals=False)
        desc = f"{task} T={T} {method}(min_N=1,max_N=50) per-DoF key-point lists (reaching.yaml's method), {uniq} distinct seeds tiled"
    elif kind == "iterative_error":        # BASELINE configs[4]: bisection on a dense synthetic A sequence
        from trajoptkp_amd import host
        uniq = min(2, B)
        dof, dt, m = synth.TASKS[task]["dof"], synth.TASKS[task]["dt"], synth.TASKS[task]["m"]
        rows, dyn = [], []
        for b in range(uniq):
            A, Bm = synth.dynamics_dense_smooth(np.random.default_rng(synth.seed_for(5, b) + 77), dof, m, dt, T)
            rows.append(host.keypoints("iterative_error", dof, T, 1, 1, iterative_error_threshold=1e-11, dt=dt, A=A)[:2]); dyn.append((A, Bm))
        rm = residuals if residuals is not None else RESIDUAL_MODEL
        p0 = synth.make_ragged_problem(task, T, rows, dyn=dyn, config_id=5, dense_residuals=rm)
        desc = f"{task} T={T} iterative_error(1e-11) ragged key-points (emulated on a dense synthetic A sequence), {uniq} distinct seeds tiled, residual Jacobians {RESIDUAL_DESC[rm]}"
    else:
        raise ValueError(kind)
    uniq = p0["batch"]
    p = synth.tile_problem(p0, B // uniq) if B > uniq else p0
    return p, p0, desc


def kp_pairs(p0):
    """Mean number of key-point (time, DoF) pairs per trajectory, and of those with DoF < num_ctrl."""
    return (float(np.mean([len(c) for (_, c) in p0["kp_rows"]])),
            float(np.mean([int(np.count_nonzero(np.asarray(c) < p0["m"])) for (_, c) in p0["kp_rows"]])))


def time_config(torch, stream, dev, p, steps, warmup, fused, generic, world=1, dist=None, kp_ordered=True, rx_const=True):
    """Times `steps` iterations of problem p on the engine; returns timings and the live engine.  rx_const: a task whose residual
    Jacobian is ONE constant matrix (reaching: p["rx_const"]) uploads it once (kpilqr_upload_residual_jacobians_const) instead
    of a copy per step -- what a host with analytic residuals does; False streams the per-step copies."""
    from trajoptkp_amd import Engine, synth
    from trajoptkp_amd import distributed as kd
    B, T = p["batch"], p["T"]
    eng = Engine(p["dof"], p["m"], T, p["nr"], batch=B, device=dev, stream=stream.cuda_stream, generic=generic,
                 fused=fused and not generic)
    is_fused = "fused" in eng.backward_variant
    tail = eng.backward_variant.rsplit("_", 1)[-1] if "tiled_" in eng.backward_variant else ""
    a4, a6 = "a4" in tail, "a6" in tail          # tiled shapes: which of a4 / a6 run inside the sweeps
    # Fused sweeps: the FD payload is resident KEY-POINT ORDERED (kpilqr_upload_fd_kp) and the backward pass differences x+ / x-
    # itself, on every timed step (one wave per trajectory: at its segment crossings; the pairs: in the helper / producer wave; per-DoF
    # lists: k_fd_kp_difference inside the backward launch sequence) -- there is no differencing stage.  With the triple
    # (KPILQR_FUSED_WAVES=4) and on materialising / tiled contexts (job lists), kpilqr_fd_difference is a stage of every timed step.
    rxc = bool(rx_const and p.get("rx_const") is not None)
    synth.upload(eng, p, kp_ordered=is_fused and kp_ordered, rx_const=rxc)
    lam = np.full(B, p["lam"])
    alphas = np.array([(i / 6.0) ** 2 for i in range(1, 7)])
    # raw: the library's own choice for a key-point ordered payload on a fused context -- the backward pass differences the
    # payload itself on EVERY call (nothing marks the column store valid).  Otherwise the differencing is a stage of its own in
    # every timed step (kpilqr_fd_difference: a backward pass alone would find the column store of the unchanged payload still
    # valid and skip it).  Asked of the library below, after the first backward pass (kpilqr_last_launch names the wave form).
    raw = False
    stages = (() if raw else ("fd_difference",)) + (() if (is_fused or a4) else ("interpolate",)) + (() if (is_fused or a6) else ("cost_derivs",)) \
        + ("backward", "forward")
    calls = {"fd_difference": eng.fd_difference, "interpolate": eng.interpolate, "cost_derivs": eng.cost_derivs,
             "backward": lambda: eng.backward(None, 100, fetch=False), "forward": lambda: eng.forward_linear(None, fetch=False)}
    if is_fused and kp_ordered:
        eng.backward(lam, 100, fetch=False)
        form = (eng.last_launch("backward").split(":") + [""])[1]
        raw = form in ("w1", "pair", "pairh") and os.environ.get("KPILQR_FUSED_RAW") != "0"      # (the forms with a raw launch sequence)
        if raw: stages = tuple(s for s in stages if s != "fd_difference")
    if "fd_difference" in stages: eng.fd_difference()
    if "interpolate" in stages: eng.interpolate()
    if "cost_derivs" in stages: eng.cost_derivs()
    eng.backward(lam, 100, fetch=False)               # uploads lambda / alphas once
    eng.forward_linear(alphas, fetch=False)
    eng.sync()
    views = None
    if world > 1:
        views = (torch.as_tensor(eng.device_array(9, (B, 6)), device="cuda"),          # KPILQR_BUF_COST_PRED
                 torch.as_tensor(eng.device_array(10, (B,)), device="cuda"),           # KPILQR_BUF_DELTA_J
                 torch.as_tensor(eng.device_array(11, (B,), "<i4"), device="cuda"))    # KPILQR_BUF_STATUS

    def one_step(events=None):
        for i, name in enumerate(stages):
            if events is not None: events[i][0].record(stream)
            calls[name]()
            if events is not None: events[i][1].record(stream)
        if world > 1:    # line-search cost reduction across GPUs: [sum_b J_pred(alpha_1..6), sum_b delta_J, #valid]
            kd.allreduce_linesearch(kd.pack_linesearch(*views))

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages] for _ in range(steps)]
    t0 = time.perf_counter()
    for s in range(steps):
        one_step(evs[s])
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stage_ms = {name: float(np.mean([evs[s][i][0].elapsed_time(evs[s][i][1]) for s in range(steps)])) for i, name in enumerate(stages)}
    return dict(eng=eng, elapsed=elapsed, stage_ms=stage_ms, stages=stages, fused=is_fused, raw=raw, kp_ordered=is_fused and kp_ordered,
                rx_const=rxc, variants={"backward": eng.backward_variant, "forward": eng.forward_variant},
                launched={"backward": eng.last_launch("backward"), "forward": eng.last_launch("forward")})


LAMBDAS = (1e-4, 1e-3, 1e-2, 0.1, 1.0, 10.0)        # the reference's range [min_lambda, max_lambda] (Optimiser.h:239-242)


def lambda_sweep(torch, stream, dev, p, fused, steps=5, rx_const=True):
    """The two sweeps of the headline batch at every regularisation of the reference's schedule and on a MIXED batch
    (trajectory b at LAMBDAS[b % 6]): the running inverse's refresh count, the LDL' re-seeds and the pivoted slow path are
    data-dependent and a launch lasts as long as its slowest wave.  Reports stage times, the number of valid backward
    passes and (when the library offers kpilqr_backward_stats) the per-step histogram of what the refresh did."""
    from trajoptkp_amd import Engine, synth
    B = p["batch"]
    eng = Engine(p["dof"], p["m"], p["T"], p["nr"], batch=B, device=dev, stream=stream.cuda_stream, fused=fused)
    synth.upload(eng, p, kp_ordered="fused" in eng.backward_variant, rx_const=rx_const)
    alphas = np.array([(i / 6.0) ** 2 for i in range(1, 7)])
    if "fused" not in eng.backward_variant:
        eng.fd_difference()
    eng.forward_linear(alphas, fetch=False)
    rows = {}
    cases = [(f"{lam:g}", np.full(B, lam)) for lam in LAMBDAS] + [("mixed", np.array([LAMBDAS[b % 6] for b in range(B)]))]
    for name, lam in cases:
        eng.backward(lam, 100, fetch=False); eng.forward_linear(None, fetch=False)     # uploads lambda; warm-up
        eng.sync()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        for s_ in range(steps):
            ev[s_][0].record(stream); eng.backward(None, 100, fetch=False)
            ev[s_][1].record(stream); eng.forward_linear(None, fetch=False)
            ev[s_][2].record(stream)
        torch.cuda.synchronize()
        bw = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])); fw = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        st = eng.results()["status"]
        row = {"stage_ms": {"backward": bw, "forward": fw}, "valid_backward_passes": int((st == 0).sum()), "batch": B}
        if hasattr(eng, "backward_stats"):
            try:
                h = eng.backward_stats(100)                                  # [B][6] step counts per trajectory
                ok = st == 0
                tot = h[ok].sum(0).astype(float)
                row["refresh_histogram"] = {"fraction_of_steps": {k: float(v / max(tot.sum(), 1.0)) for k, v in zip(
                    ("third_order_only", "plus_1_second_order", "plus_2_second_order", "plus_3_second_order", "ldl_factorisation", "pivoted_slow_path"), tot)},
                    "slowest_trajectory_extra_refresh_steps": int((h[:, 1] + 2 * h[:, 2] + 3 * h[:, 3]).max())}
            except Exception as ex:
                row["refresh_histogram"] = {"error": repr(ex)}
        rows[name] = row
    eng.close()
    return rows


def roofline_of(p, p0, r, pmc=None):
    """roofline object of the backward sweep of a timed configuration."""
    n, m, nr, T, B = p["n"], p["m"], p["nr"], p["T"], p["batch"]
    t_bwd = r["stage_ms"]["backward"] * 1e-3
    Kp_steps = float(np.mean([np.count_nonzero(np.diff(o)) for (o, _) in p0["kp_rows"]]))
    ab = algorithmic_bytes(p["dof"], m, nr, T, Kp_steps, 6)
    var = r["variants"]["backward"]
    tail = var.rsplit("_", 1)[-1] if "tiled_" in var else ""
    a4, a6 = r["fused"] or "a4" in tail, r["fused"] or "a6" in tail
    pairs = kp_pairs(p0)
    kb = 8 * T * (m * n + m)                                                        # gains out
    cols = 8 * (pairs[0] * 2 * n + pairs[1] * n)                                    # the key-point columns of a trajectory
    # A, B: key-point columns (the raw sweep reads x+ and x- and writes the differenced column: 3x) or every step
    kb += (3 * cols + pairs[0] if r.get("raw") else cols) if a4 else 8 * T * (n * n + n * m)
    ru0 = r["fused"] and not np.any(p0["r_u"])            # r_u never uploaded: the fused backward sweep does not read it
    rxc = ":rxc" in r.get("launched", {}).get("backward", "")      # the constant Jacobian sits in registers: no r_x traffic
    kb += 8 * T * nr * (1 + (0 if rxc else n) + (0 if ru0 else m)) if a6 else 8 * T * (n * n + n + m * m + m)   # residuals + Jacobians or l_*
    flops = flops_a7(n, m) * T * B
    ach_tf = flops / t_bwd / 1e12
    traffic = None
    rxc_pmc = ":rxc" in r.get("launched", {}).get("backward", "")
    if pmc is not None:
        try:
            wl = pmc["workload"]
            if wl["task"] == p["task"] and wl["T"] == T and wl["batch"] == B and bool(wl.get("rx_const", False)) == rxc_pmc:
                traffic = pmc["kernels"]["backward_fused" if r["fused"] else "backward"]["traffic_bytes"]
        except Exception:
            traffic = None
    hbm = {"kernel_compulsory_bytes_per_launch": kb * B, "achieved_GBps": kb * B / t_bwd / 1e9,
           "frac_of_hbm_peak": kb * B / t_bwd / 1e9 / HBM_PEAK_GBS,
           "survey_8d_riccati_bytes_per_launch": ab["backward"] * B,
           "survey_8d_GBps": ab["backward"] * B / t_bwd / 1e9, "survey_8d_frac_of_hbm_peak": ab["backward"] * B / t_bwd / 1e9 / HBM_PEAK_GBS,
           "peak_GBps": HBM_PEAK_GBS}
    if traffic is not None:
        hbm["traffic_GBps"] = traffic / t_bwd / 1e9
        hbm["traffic_frac_of_hbm_peak"] = traffic / t_bwd / 1e9 / HBM_PEAK_GBS
    return {"bound": "mfma", "kernel": f"backward ({r.get('launched', r['variants'])['backward']})", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
            "traffic_source": (None if traffic is None else f"profiles/{pmc.get('_file', '?')}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                               "passes of this command (tools/collect_profiles.sh), NOT measured in this run"),
            "traffic_source_short": (None if traffic is None else f"profiles/{pmc.get('_file', '?')} (separate rocprofv3 --pmc passes; not measured in this run)"),
            "avg_launch_ms": r["stage_ms"]["backward"],
            "algorithmic_flops_per_launch": flops, "flops_per_trajectory_step": flops_a7(n, m), "hbm": hbm}


def parity_check(p0, eng, n_check, tiled=False):
    """K, k, delta_J, cost_pred of the first n_check trajectories against the CPU oracle (the timed engine's last results);
    tiled: the batch is p0 replicated, and every replica must carry its seed's bytes."""
    from oracle import pipeline
    K, k = eng.gains()
    res = eng.results()
    out = {"trajectories_checked": n_check, "max_rel_err_K": 0.0, "max_rel_err_k": 0.0, "max_rel_err_cost_pred": 0.0,
           "max_rel_err_delta_J": 0.0, "tolerance_K": 1e-6, "oracle": "oracle/kpilqr_oracle.c (parity unpinned for a6-a9: SURVEY 8c)"}
    for b in range(n_check):
        o = pipeline.run_trajectory(p0, b)
        rel = lambda a, r_: float(np.max(np.abs(a - r_)) / max(float(np.max(np.abs(r_))), 1e-300))
        out["max_rel_err_K"] = max(out["max_rel_err_K"], rel(K[b], o["K"]))
        out["max_rel_err_k"] = max(out["max_rel_err_k"], rel(k[b], o["k"]))
        out["max_rel_err_cost_pred"] = max(out["max_rel_err_cost_pred"], rel(res["cost_pred"][b], o["cost_pred"]))
        out["max_rel_err_delta_J"] = max(out["max_rel_err_delta_J"], abs(res["delta_J"][b] - o["delta_J"]) / abs(o["delta_J"]))
        if o["status"] != 0 or res["status"][b] != 0:
            out["status_mismatch"] = True
    uniq = p0["batch"]
    reps = K.shape[0] // uniq if tiled else 1
    if reps > 1:        # every replica of a seed must carry its seed's bytes
        Kr = K.reshape(reps, uniq, -1)
        out["replicas_bit_identical"] = bool(np.array_equal(Kr, np.broadcast_to(Kr[0], Kr.shape)))
    out["pass"] = bool(out["max_rel_err_K"] < 1e-6 and out["max_rel_err_k"] < 1e-6 and out["max_rel_err_cost_pred"] < 1e-6
                       and out["max_rel_err_delta_J"] < 1e-6 and not out.get("status_mismatch", False)
                       and out.get("replicas_bit_identical", True))
    return out


def pcie_inclusive(batch, steps=4):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pcie_inclusive as pi
    m = pi.measure(batch, steps, chunk_list=(3,), quiet=True)
    pick = lambda payload, form: next(r for r in m["rows"] if (r["payload"] == payload or r["payload"].startswith(payload + " (")) and r["form"].startswith(form))
    out = {"batch": m["batch"], "unit": "trajectory-iterations/s", "resident_value": m["resident_traj_it_per_s"],
           "note": "H2D of the FD payload + residuals (+ Jacobians) and D2H of K,k every iteration, pinned host memory; "
                   "kpilqr_iterate_streamed over 3 trajectory chunks (SDMA uploads | kernels | kernel downloads overlapped); "
                   "'pipelined' = consecutive iterations enqueued without a host wait, 'synced' = host waits after every iteration; "
                   "'serial' = round-1 call sequence (upload_fd + upload_residuals + iterate + download_gains + sync)"}
    for key, payload in (("full_payload", "full payload"), ("resident_jacobians", "resident Jacobians")):
        a, b, c = pick(payload, "chunks=3 pipelined"), pick(payload, "chunks=3 per-iteration"), pick(payload, "serial")
        out[key] = {"value": a["traj_it_per_s"], "ms_per_iteration": a["ms_per_iteration"], "h2d_GB": a["h2d_GB"], "d2h_GB": a["d2h_GB"],
                    "link_GBps": a["link_GBps"], "synced_value": b["traj_it_per_s"], "serial_value": c["traj_it_per_s"]}
    # the same with the key-point columns differenced on the host (kpilqr_upload_kp_columns): half the FD bytes, bit-identical K
    for key, payload in (("full_payload_host_differenced_columns", "columns + full residual payload"),
                         ("resident_jacobians_host_differenced_columns", "columns + resident Jacobians")):
        a = pick(payload, "chunks=3 pipelined")
        out[key] = {"value": a["traj_it_per_s"], "ms_per_iteration": a["ms_per_iteration"], "h2d_GB": a["h2d_GB"], "d2h_GB": a["d2h_GB"],
                    "link_GBps": a["link_GBps"]}
    # round 4: ONE constant residual Jacobian uploaded once (kpilqr_upload_residual_jacobians_const) -- the full per-iteration
    # payload of a task with analytic residuals is then the FD payload (or the host-differenced columns) + the residuals
    for key, payload in (("full_payload_constant_jacobians", "constant Jacobians"), ("full_payload_constant_jacobians_host_differenced_columns", "columns + constant Jacobians")):
        try:
            a = pick(payload, "chunks=3 pipelined")
            out[key] = {"value": a["traj_it_per_s"], "ms_per_iteration": a["ms_per_iteration"], "h2d_GB": a["h2d_GB"], "d2h_GB": a["d2h_GB"],
                        "link_GBps": a["link_GBps"]}
        except StopIteration:
            pass
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), as a CHILD process --
    this process has not touched the GPU and never will -- pass rank 0's JSON line through and exit with the child's code.
    With fewer GPUs than ranks (rehearsal on a one-GPU box) the ranks share devices: RCCL refuses a communicator with two
    ranks on one device, so the rehearsal's collective runs over gloo and the line says so (`rccl_ranks` = 0)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()          # does not initialise HIP
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < n:
        env.setdefault("KPILQR_DIST_BACKEND", "gloo")
        print(f"bench.py: {n} ranks on {ndev} GPU(s): rehearsal, ranks share devices, collective over gloo", file=sys.stderr)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    # rank 0's JSON line goes to stdout, whatever else the ranks print there (gloo's connection banner in a rehearsal) to stderr
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=1024, help="trajectories of the whole job, sharded over the ranks (strong scaling)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --batch trajectories PER GPU")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (implies --weak for N>1; at N=1 the same as --global-batch)")
    ap.add_argument("--T", type=int, default=3000)
    ap.add_argument("--min-N", type=int, default=5)
    ap.add_argument("--task", default="panda_reaching")
    ap.add_argument("--keypoints", default="set_interval", choices=["set_interval", "adaptive_jerk", "iterative_error", "reach_velocity_change", "reach_adaptive_jerk"])
    ap.add_argument("--generic", action="store_true", help="force the generic (VALU/LDS, non-MFMA) kernels")
    ap.add_argument("--unfused", action="store_true", help="materialise A,B (interpolate) and l_* (cost_derivs) with their own kernels")
    ap.add_argument("--no-secondary", action="store_true", help="skip every side measurement (materialising pipeline, pcie_inclusive, secondary configs, weak line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie-batch", type=int, default=256)
    ap.add_argument("--workload-cache", default=None, help="directory for the generated workload (.npy, memory-mapped by later runs: profiler passes)")
    ap.add_argument("--tiled-seeds", action="store_true", help="8 distinct seeds tiled to the batch (round-1/2 workload) instead of one seed per trajectory")
    ap.add_argument("--detail", default=None, help="path of the detail file (default: bench_detail.json beside this script)")
    ap.add_argument("--streamed-jacobians", action="store_true",
                    help="upload the residual Jacobians per step (T+1 copies per trajectory) even when the task has ONE constant matrix "
                         "(reaching): the round-1..3 form; default since round 4 is kpilqr_upload_residual_jacobians_const")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)", file=sys.stderr)
        sys.exit(2)

    from trajoptkp_amd import distributed as kd
    weak = args.weak or (args.batch is not None and world > 1)
    if weak:
        B_local = args.batch or 1024
        global_batch = B_local * world
        lo = rank * B_local
    else:
        global_batch = args.batch if (args.batch is not None and world == 1) else args.global_batch
        lo, hi = kd.shard_range(global_batch, rank, world)          # contiguous block of the global batch
        B_local = hi - lo
    T = args.T
    # The workload is generated BEFORE this process touches the GPU: the distinct seeds are made by forked workers.
    # Every rank makes the seeds of its own shard (trajectory lo + b has seed seed_for(2, lo + b), whatever N is).
    t_gen = time.perf_counter()
    p, p0, desc = build_problem(args.keypoints, B_local, T, args.min_N, args.task, first_b=lo, distinct=not args.tiled_seeds,
                                cache=args.workload_cache or os.environ.get("KPILQR_WORKLOAD_CACHE"))
    pw = None
    if world > 1 and not weak and not args.no_secondary:            # the weak-scaling line beside the strong one
        pw = build_problem(args.keypoints, args.global_batch, T, args.min_N, args.task, distinct=False)[0]   # 8 seeds tiled: a side line
    t_gen = time.perf_counter() - t_gen

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()     # the modulo only matters when rehearsing N>1 ranks on a 1-GPU box
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("KPILQR_DIST_BACKEND", "nccl")        # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # a dedicated (non-null) HIP stream shared by torch and the engine: kernels, HIP events and the RCCL all-reduce
    # are all ordered on it
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    fused = not args.unfused and not args.generic

    rx_const = not args.streamed_jacobians
    r = time_config(torch, stream, local_rank, p, args.steps, args.warmup, fused, args.generic, world, dist, rx_const=rx_const)
    eng = r["eng"]
    res = eng.results()
    n_ok = int((res["status"] == 0).sum())
    tiled = args.tiled_seeds or args.keypoints != "set_interval"
    parity = parity_check(p0, eng, min(p0["batch"], 8), tiled) if rank == 0 else None
    eng.close()

    side = world == 1 and not args.no_secondary
    weak_line = None
    if pw is not None:
        # the same job with 1024 trajectories PER GPU, for the weak-scaling curve beside the strong one
        rw = time_config(torch, stream, local_rank, pw, max(3, args.steps // 2), 2, fused, args.generic, world, dist, rx_const=rx_const)
        rw["eng"].close()
        weak_line = {"batch_per_gpu": args.global_batch, "global_batch": args.global_batch * world, "steps": max(3, args.steps // 2),
                     "value": args.global_batch * world * max(3, args.steps // 2) / rw["elapsed"], "unit": "trajectory-iterations/s",
                     "ms_per_step": 1e3 * rw["elapsed"] / max(3, args.steps // 2)}

    if rank == 0:
        value = global_batch * args.steps / r["elapsed"]
        pmc = None
        for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                pmc["_file"] = name
                break
            except Exception:
                pmc = None
        pmc3 = None           # the streamed-Jacobian form: PMC passes of round 3 (or a round-4 pass with --streamed-jacobians)
        for name in ("r05_pmc_traffic_per_step_jacobians.json", "r04_pmc_traffic_per_step_jacobians.json", "r03_pmc_traffic.json"):
            try:
                pmc3 = json.load(open(os.path.join(ROOT, "profiles", name))); pmc3["_file"] = name
                break
            except Exception:
                pmc3 = None
        ab = algorithmic_bytes(p["dof"], p["m"], p["nr"], T, float(np.mean([np.count_nonzero(np.diff(o)) for (o, _) in p0["kp_rows"]])), 6)
        roof = roofline_of(p, p0, r, (pmc if r.get("rx_const") else pmc3) if not args.generic else None)
        out = {
            "metric": "iLQR iterations/sec (Panda 7-DoF, T=3000)" if args.task == "panda_reaching" and T == 3000
                      else f"iLQR iterations/sec ({args.task}, T={T})",
            "value": value, "unit": "trajectory-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * r["elapsed"] / args.steps, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{desc} per rank, global batch {global_batch} sharded over {world} GPU(s) ({B_local} on rank 0), "
                                   f"6 alphas, lambda={p['lam']}"
                                   + (", fused sweeps (a4+a6 inside a7/a8)" if r["fused"] else "")
                                   + (", ONE constant residual Jacobian uploaded once (the task's r_x: Reaching.cpp:43-54) and kept in registers by the sweeps" if r.get("rx_const") and ":rxc" in r["launched"]["backward"]
                                      else ", constant residual Jacobian uploaded once, read by the sweeps from its broadcast copy" if r.get("rx_const") else ", residual Jacobians streamed per step")
                                   + (", key-point ordered FD payload differenced inside the backward sweep (no fd_difference stage)" if r.get("raw") else
                                      ", key-point ordered FD payload" if r.get("kp_ordered") else ""),
                       "workload_short": f"{args.task} T={T} {args.keypoints}({args.min_N}), {'8 seeds tiled' if args.tiled_seeds or args.keypoints != 'set_interval' else 'distinct seeds'}, "
                                         f"global batch {global_batch} over {world} GPU(s), 6 alphas, lambda={p['lam']}",
                       "batch_per_gpu": B_local, "global_batch": global_batch, "horizon": T, "kernels": r["variants"], "launched": r["launched"],
                       "residual_jacobians": "constant" if r.get("rx_const") else "per_step",
                       "residual_jacobians_source": ("ONE r_x uploaded once (kpilqr_upload_residual_jacobians_const), as host/iLQR_GPU does when the task's r_x is constant"
                                                     if r.get("rx_const") else "r_x, r_u uploaded per step (kpilqr_upload_residuals)"),
                       "valid_backward_passes_rank0": n_ok, "parallelism": f"traj-shard x{world}",
                       "workload_generation_s": t_gen},
            "rccl_ranks": (dist.get_world_size() if (world > 1 and dist.get_backend() == "nccl") else (1 if world == 1 else 0)),
            "collective": ("none (one rank)" if world == 1 else
                           f"one all-reduce of 8 doubles per iteration, torch.distributed '{dist.get_backend()}'"
                           + (" (= RCCL)" if dist.get_backend() == "nccl" else " (REHEARSAL: ranks share GPUs, RCCL needs a device per rank)")),
            "batch_iterations_per_s": args.steps / r["elapsed"],
            "stage_ms": r["stage_ms"],
            "roofline": roof,
            # SURVEY 8(d)'s whole-iteration figure (A, B, l_* counted as if materialised): comparable across rounds, NOT what
            # the fused pipeline moves
            "survey_8d_iteration": {"bytes_per_trajectory": sum(ab.values()),
                                    "GBps": sum(ab.values()) * B_local / (r["elapsed"] / args.steps) / 1e9,
                                    "frac_of_hbm_peak": sum(ab.values()) * B_local / (r["elapsed"] / args.steps) / 1e9 / HBM_PEAK_GBS},
            "parity_check": parity,
        }
        # issued (not algorithmic) FP64 MFMA work of the backward sweep, from the PMC pass of the same workload
        try:
            if r["fused"] and args.task == "panda_reaching" and not args.generic:
                cnt = None
                for name in ("r05_pmc_counters.json", "r04_pmc_counters.json", "r03_pmc_counters.json", "r02_pmc_counters.json", "r01_pmc_counters.json"):
                    try:
                        cnt = json.load(open(os.path.join(ROOT, "profiles", name)))["derived"]["backward_fused"]; break
                    except Exception:
                        cnt = None
                issued = cnt["mfma_per_step_per_trajectory"] * 2048.0 * T * B_local
                ach_tf = issued / (r["stage_ms"]["backward"] * 1e-3) / 1e12
                out["roofline"]["issued_mfma"] = {"TFLOPs": ach_tf, "frac_of_fp64_peak": ach_tf / FP64_PEAK_TFLOPS,
                                                  "mfma_16x16x4_per_trajectory_step": cnt["mfma_per_step_per_trajectory"],
                                                  "source": f"profiles/{name} (a separate rocprofv3 --pmc pass, not measured in this run)"}
        except Exception:
            pass
        if weak_line is not None:
            out["weak_scaling"] = weak_line
        if side:
            # ---- the materialising five-kernel pipeline beside the fused default ------------------------------------
            if r["fused"]:
                r2 = time_config(torch, stream, local_rank, p, max(3, min(args.steps, 10)), 2, False, False)
                r2["eng"].close()
                k2 = max(3, min(args.steps, 10))
                out["materialising_pipeline"] = {"value": B_local * k2 / r2["elapsed"], "unit": "trajectory-iterations/s", "steps": k2,
                                                 "ms_per_step": 1e3 * r2["elapsed"] / k2, "kernels": r2["variants"], "stage_ms": r2["stage_ms"],
                                                 "stage_algorithmic_GBps": {k: ab[k] * B_local / (r2["stage_ms"][k] * 1e-3) / 1e9 for k in r2["stages"]},
                                                 "roofline": roofline_of(p, p0, r2)}
            # ---- the same iteration with the residual Jacobians given per step (rounds 1-3; what a task with state-dependent
            # residual Jacobians pays) ------------------------------------------------------------------------------------------
            if r.get("rx_const"):
                try:
                    k3 = max(3, min(args.steps, 10))
                    r3 = time_config(torch, stream, local_rank, p, k3, 2, fused, args.generic, rx_const=False)
                    r3["eng"].close()
                    out["per_step_residual_jacobians"] = {"value": B_local * k3 / r3["elapsed"], "unit": "trajectory-iterations/s", "steps": k3,
                                                          "ms_per_step": 1e3 * r3["elapsed"] / k3, "stage_ms": r3["stage_ms"], "launched": r3["launched"],
                                                          "roofline": roofline_of(p, p0, r3, pmc3)}
                except Exception as ex:
                    out["per_step_residual_jacobians"] = {"error": repr(ex)}
            # ---- the regularisation range and a mixed batch ------------------------------------------------------------
            if args.task == "panda_reaching" and not args.generic:
                try:
                    out["lambda_sweep"] = lambda_sweep(torch, stream, local_rank, p, fused, rx_const=rx_const)
                    if r.get("raw"):
                        # the kernel as launched also differences the FD payload (a2); the same sweep on a column store that
                        # is already differenced (what the lambda sweep times) is the a7 kernel proper
                        ms = out["lambda_sweep"][f"{p['lam']:g}"]["stage_ms"]["backward"]
                        tf = out["roofline"]["algorithmic_flops_per_launch"] / (ms * 1e-3) / 1e12
                        out["roofline"]["sweep_on_differenced_columns"] = {"avg_launch_ms": ms, "achieved": tf, "frac": tf / FP64_PEAK_TFLOPS,
                                                                           "note": "k_backward_fused_excl<..., RAW = false>: no differencing inside"}
                except Exception as ex:
                    out["lambda_sweep"] = {"error": repr(ex)}
            # ---- PCIe-inclusive rate -------------------------------------------------------------------
            if args.task == "panda_reaching" and args.keypoints == "set_interval":
                try:
                    out["pcie_inclusive"] = pcie_inclusive(args.pcie_batch)
                except Exception as ex:
                    out["pcie_inclusive"] = {"error": repr(ex)}
                if B_local >= 1024 and args.pcie_batch != 1024:     # the headline batch too (13 GB of pinned host memory)
                    try:
                        out["pcie_inclusive_b1024"] = pcie_inclusive(1024, steps=2)
                    except Exception as ex:
                        out["pcie_inclusive_b1024"] = {"error": repr(ex)}
            # ---- BASELINE configs[1], [2], [4] ----------------------------------------------------------------------
            if args.task == "panda_reaching" and args.keypoints == "set_interval" and T == 3000 and not args.generic and not args.unfused:
                sec = {}
                # (configs[2] and [4] twice: with residual Jacobians drawn independently at every step -- the workload of rounds 1-3,
                # on which the running inverse of Q_uu never applies -- and with Jacobians of the reference's structure, smooth in time)
                for key, (kind, task, Ts, Bs, ks, rm) in {
                        "configs[1] panda_reaching T=3000 batch=1": ("set_interval", "panda_reaching", 3000, 1, 10, None),
                        # the headline shape with the reference's OWN key-point method for reaching (reaching.yaml:6-8): per-DoF lists
                        "panda_reaching T=3000 velocity_change(1,50) batch=1024 (reaching.yaml's key-point method)": ("reach_velocity_change", "panda_reaching", 3000, 1024, 8, None),
                        "panda_reaching T=3000 adaptive_jerk(1,50) batch=1024": ("reach_adaptive_jerk", "panda_reaching", 3000, 1024, 8, None),
                        "configs[2] panda_pushing T=3000 adaptive_jerk batch=64": ("adaptive_jerk", "panda_pushing", 3000, 64, 5, True),
                        "configs[2] panda_pushing T=3000 adaptive_jerk batch=64, smooth residual Jacobians": ("adaptive_jerk", "panda_pushing", 3000, 64, 5, "smooth"),
                        "configs[4] high_dof_push n=62 T=5000 iterative_error batch=128 (one GPU's share of 1024)": ("iterative_error", "high_dof_push", 5000, 128, 3, True),
                        "configs[4] high_dof_push n=62 T=5000 iterative_error batch=128 (one GPU's share of 1024), smooth residual Jacobians": ("iterative_error", "high_dof_push", 5000, 128, 3, "smooth"),
                        # not a BASELINE config: the reference's humanoid (TaskConfigs/locomotion/humanoid.yaml, 21 actuators) on the
                        # wide-control tiled sweeps (tiled_wide.hip) instead of the VALU / LDS kernels
                        "humanoid n=54 m=21 T=1500 set_interval batch=64": ("set_interval", "humanoid", 1500, 64, 3, None)}.items():
                    try:
                        ps, ps0, ds = build_problem(kind, Bs, Ts, 5, task, distinct=False, residuals=rm)
                        rs = time_config(torch, stream, local_rank, ps, ks, 1, True, False)
                        pc = parity_check(ps0, rs["eng"], min(ps0["batch"], 2), tiled=kind != "set_interval")
                        rs["eng"].close()
                        sec[key] = {"workload": ds + f", batch={Bs}", "value": Bs * ks / rs["elapsed"], "unit": "trajectory-iterations/s",
                                    "steps": ks, "ms_per_step": 1e3 * rs["elapsed"] / ks, "kernels": rs["variants"], "launched": rs["launched"], "stage_ms": rs["stage_ms"],
                                    "keypoint_pairs_per_trajectory": kp_pairs(ps0)[0], "roofline": roofline_of(ps, ps0, rs),
                                    "parity_check": {k: pc[k] for k in ("max_rel_err_K", "max_rel_err_cost_pred", "pass")}}
                        # the CPU oracle on the same workload: threads = min(batch, this box's CPU share)
                        if not args.no_cpu_baseline and (key.startswith("configs[") or "velocity_change" in key):
                            try:
                                cb = cpu_baseline_problem(ps0, Bs, budget_s=4.0 if "smooth" in key else 8.0)
                                sec[key]["cpu_baseline"] = cb
                                sec[key]["gpu_over_cpu"] = sec[key]["value"] / cb["value"]
                                sec[key]["meets_50x"] = bool(sec[key]["gpu_over_cpu"] >= 50.0)
                            except Exception as ex:
                                sec[key]["cpu_baseline"] = {"error": repr(ex)}
                        del ps, ps0, rs
                    except Exception as ex:
                        sec[key] = {"error": repr(ex)}
                out["secondary_configs"] = sec
                # ---- BASELINE configs[3] on ONE GPU: what a shard of the global batch of 1024 costs here.  A projection, labelled as
                # one: N x (the rate of a 1024/N shard measured on this GPU); the driver's --gpus N runs measure the real thing.
                proj = {}
                for N_ in (2, 4, 8):
                    try:
                        Bs = B_local // N_
                        ps = slice_problem(p, Bs)                       # the first 1024/N of the headline batch's trajectories
                        rs = time_config(torch, stream, local_rank, ps, 10, 2, True, False)
                        rs["eng"].close()
                        per = Bs * 10 / rs["elapsed"]
                        proj[str(N_)] = {"batch_per_gpu": Bs, "ms_per_step": 1e3 * rs["elapsed"] / 10, "per_gpu_value": per,
                                         "projected_value": N_ * per, "stage_ms": rs["stage_ms"]}
                        del ps, rs
                    except Exception as ex:
                        proj[str(N_)] = {"error": repr(ex)}
                out["weak_scaling_projection"] = {"note": f"{B_local} trajectories PER GPU (what `bench.py --gpus N --weak` runs): N x this GPU's measured rate -- shards share "
                                                          "nothing and the only collective is one 64-byte all-reduce per iteration; a projection, not a measurement",
                                                  "unit": "trajectory-iterations/s",
                                                  "n_gpus": {str(N_): {"batch_per_gpu": B_local, "global_batch": N_ * B_local, "projected_value": N_ * value} for N_ in (1, 2, 4, 8)}}
                out["strong_scaling_projection"] = {"note": "global batch 1024 over N GPUs, projected from shards timed on THIS GPU (no "
                                                            "inter-GPU cost: the only collective is one 64-byte all-reduce per iteration)",
                                                    "unit": "trajectory-iterations/s", "n_gpus": proj}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task, T, args.min_N)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
                # a single trajectory cannot use 16 CPU threads either: the latency comparison for --batch 1
                if out["cpu_baseline"].get("single_thread_value"):
                    out["gpu_over_cpu_single_thread"] = value / out["cpu_baseline"]["single_thread_value"]
            except Exception as ex:   # the baseline is reporting only; never hide the GPU number
                out["cpu_baseline"] = {"error": repr(ex)}
            # which BASELINE configs reach the north star's ">= 50x the CPU reference at 1 GPU" -- said plainly
            try:
                bar = {"configs[3] (headline, this line)": [out.get("gpu_over_cpu"), bool(out.get("gpu_over_cpu", 0) >= 50.0)]}
                for k_, v_ in (out.get("secondary_configs") or {}).items():
                    if isinstance(v_, dict) and "gpu_over_cpu" in v_:
                        bar[k_] = [v_["gpu_over_cpu"], v_["meets_50x"]]
                out["gpu_over_cpu_by_config"] = bar
            except Exception:
                pass
        emit(out, args.detail or os.path.join(ROOT, "bench_detail.json"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
