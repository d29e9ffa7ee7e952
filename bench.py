#!/usr/bin/env python3
"""bench.py -- iLQR iterations/sec on the keypoint-iLQR hot path (BASELINE.json metric).

One "step" = one whole GPU-side iLQR iteration for every trajectory of the batch:
    fd_difference (a2) -> interpolate (a4) -> cost_derivs (a6) -> backward pass (a7, one pass at a
    valid lambda) -> linearised forward pass over the 6 line-search alphas (a8)
By default a4 and a6 run INSIDE the two sweeps (KPILQR_FLAG_FUSED, trajoptkp_amd/csrc/fused_mfma.hip:
three launches per iteration, A/B/l_* never written to HBM); `--unfused` times the materialising
five-kernel pipeline instead, and at N=1 the default run reports it beside the headline number
("materialising_pipeline").  All inputs (host FD results, residuals and their Jacobians, nominal controls) already resident
in HBM when the timed region starts.  Workload (N=1 and per rank for N>1, weak scaling): Franka Panda
7-DoF reaching, T=3000, set-interval key-points every 5 steps, batch=1024 independent trajectories
(BASELINE configs[3]; configs[1] is the same problem at batch=1: `--batch 1`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel = backward pass, HIP-event
timed on the launch stream) and "cpu_baseline" (the CPU oracle = line-faithful port of the
reference, timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(dof, m, nr, T, Kp, n_alpha):
    """compulsory FP64 stage I/O per trajectory-iteration."""
    n = 2 * dof
    return dict(
        fd_difference=8 * Kp * ((2 * dof + m) * 2 * n + (n * n + n * m)),
        interpolate=8 * (Kp * (n * n + n * m) + T * (n * n + n * m)),
        cost_derivs=8 * ((T + 1) * nr * (1 + n + m) + T * (n + n * n + m + m * m)),
        backward=8 * T * ((n * n + n * m + n + n * n + m + m * m) + (m * n + m)),
        forward=8 * T * (n * n + n * m + m * n + m + n + n * n + m + m * m) + 8 * n_alpha,
    )


def cpu_baseline(task, T, min_N, reps_per_thread=60):
    """Times the CPU oracle (oracle/kpilqr_oracle.c = line-faithful port of the reference, built here with
    -O3 -march=native) on this host: whole trajectory-iterations (the same five stages) run by a pthread
    pool inside the C library, one independent trajectory per thread at a time -- the batch analogue of
    the reference's hardware_concurrency() thread pools.  Thread count = the 1-GPU box's CPU share (16)."""
    import tempfile
    from oracle import oracle as orc
    from trajoptkp_amd import synth
    path = orc.build(native=True, out_dir=tempfile.mkdtemp(prefix="kpilqr_oracle_"))
    orc._LIB = orc.lib(path)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("KPILQR_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores
    p = synth.make_problem(task=task, T=T, batch=1, min_N=min_N)
    orc.iteration_batch_seconds(p, 0, 1, 1)                                # warm-up
    t_single = orc.iteration_batch_seconds(p, 0, 1, 5) / 5
    wall = orc.iteration_batch_seconds(p, 0, cores, reps_per_thread)
    n_traj = cores * reps_per_thread
    return {"value": n_traj / wall, "unit": "trajectory-iterations/s", "cores": cores, "kind": "port",
            "sample": f"{n_traj} trajectory-iterations ({task}, T={T}, key-points every {min_N}) on {cores} pthreads "
                      f"({wall:.1f} s wall); single thread: {1.0 / t_single:.2f} it/s",
            "single_thread_value": 1.0 / t_single}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU")
    ap.add_argument("--T", type=int, default=3000)
    ap.add_argument("--min-N", type=int, default=5)
    ap.add_argument("--task", default="panda_reaching")
    ap.add_argument("--unique", type=int, default=8, help="distinct seeded trajectories, tiled to --batch")
    ap.add_argument("--generic", action="store_true", help="force the generic (non-MFMA) kernels")
    ap.add_argument("--unfused", action="store_true",
                    help="materialise A,B (interpolate) and l_* (cost_derivs) with their own kernels instead of "
                         "evaluating them inside the sweeps (KPILQR_FLAG_FUSED)")
    ap.add_argument("--fused", action="store_true", help="(default) fused sweeps: a4+a6 inside a7/a8")
    ap.add_argument("--no-secondary", action="store_true", help="skip the materialising-pipeline side measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from trajoptkp_amd import Engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; the modulo only matters when rehearsing N>1 ranks on a 1-GPU box (gloo backend)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("KPILQR_DIST_BACKEND", "nccl")        # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    B, T = args.batch, args.T
    uniq = min(args.unique, B)
    reps = (B + uniq - 1) // uniq
    p = synth.make_problem(task=args.task, T=T, batch=uniq, min_N=args.min_N, first_b=rank * uniq)
    if reps > 1:
        p = synth.tile_problem(p, reps)
    B = p["batch"]
    # a dedicated (non-null) HIP stream shared by torch and the engine: kernels, HIP events and the
    # RCCL all-reduce are all ordered on it
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng = Engine(p["dof"], p["m"], T, p["nr"], batch=B, device=local_rank, stream=stream.cuda_stream,
                 generic=args.generic,
                 # fused sweeps at every batch size: wave pairs per trajectory up to #SIMDs/2 trajectories, one wave
                 # per trajectory beyond (tools/small_batch_variants.sh: B=1 161 vs 143 it/s materialising)
                 fused=not args.unfused and not args.generic)
    fused = "fused" in eng.backward_variant
    synth.upload(eng, p)
    lam = np.full(B, p["lam"])
    alphas = np.array([(i / 6.0) ** 2 for i in range(1, 7)])
    eng.fd_difference()
    eng.backward(lam, 100, fetch=False)               # uploads lambda / alphas once
    eng.forward_linear(alphas, fetch=False)
    eng.sync()
    Kp = len(p["kp_times"])
    ab = algorithmic_bytes(p["dof"], p["m"], p["nr"], T, Kp, 6)
    if fused:
        # compulsory I/O of the fused sweeps: key-point columns + residuals/Jacobians in, gains out (backward);
        # the same plus gains and nominal controls in (forward)
        n_, m_, nr_ = p["n"], p["m"], p["nr"]
        src = 8 * (Kp * (n_ * n_ + n_ * m_) + T * nr_ * (1 + n_ + m_))
        ab_fused = dict(fd_difference=ab["fd_difference"], backward=src + 8 * T * (m_ * n_ + m_),
                        forward=src + 8 * T * (m_ * n_ + m_ + m_) + 8 * 6)

    # line-search cost reduction across GPUs: [sum_b J_pred(alpha_1..6), sum_b delta_J, #valid] (8 doubles)
    from trajoptkp_amd import distributed as kd
    cost_view = torch.as_tensor(eng.device_array(9, (B, 6)), device="cuda")       # KPILQR_BUF_COST_PRED
    dJ_view = torch.as_tensor(eng.device_array(10, (B,)), device="cuda")          # KPILQR_BUF_DELTA_J
    st_view = torch.as_tensor(eng.device_array(11, (B,), "<i4"), device="cuda")   # KPILQR_BUF_STATUS

    a6 = eng.backward_variant.endswith("_a6")         # tiled shapes: a6 inside the sweeps, A and B still materialised
    stages = ("fd_difference", "backward", "forward") if fused else \
             ("fd_difference", "interpolate", "backward", "forward") if a6 else \
             ("fd_difference", "interpolate", "cost_derivs", "backward", "forward")

    def one_step(events=None):
        for i, name in enumerate(stages):
            if events is not None:
                events[i][0].record(stream)
            if name == "fd_difference": eng.fd_difference()
            elif name == "interpolate": eng.interpolate()
            elif name == "cost_derivs": eng.cost_derivs()
            elif name == "backward": eng.backward(None, 100, fetch=False)      # lambda stays resident
            else: eng.forward_linear(None, fetch=False)                       # alphas stay resident
            if events is not None:
                events[i][1].record(stream)
        if world > 1:
            kd.allreduce_linesearch(kd.pack_linesearch(cost_view, dJ_view, st_view))

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages]
           for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(evs[s])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stage_ms = {name: float(np.mean([evs[s][i][0].elapsed_time(evs[s][i][1]) for s in range(args.steps)]))
                for i, name in enumerate(stages)}
    res = eng.results()
    n_ok = int((res["status"] == 0).sum())
    variants = {"backward": eng.backward_variant, "forward": eng.forward_variant}

    # side measurement (N=1 only, outside the timed region above): the materialising five-kernel pipeline
    secondary = None
    if fused and world == 1 and not args.no_secondary:
        eng.close()
        eng = Engine(p["dof"], p["m"], T, p["nr"], batch=B, device=local_rank, stream=stream.cuda_stream, fused=False)
        synth.upload(eng, p)
        eng.fd_difference(); eng.interpolate(); eng.cost_derivs()
        eng.backward(lam, 100, fetch=False); eng.forward_linear(alphas, fetch=False); eng.sync()
        st2 = ("fd_difference", "interpolate", "cost_derivs", "backward", "forward")
        calls = {"fd_difference": eng.fd_difference, "interpolate": eng.interpolate, "cost_derivs": eng.cost_derivs,
                 "backward": lambda: eng.backward(None, 100, fetch=False), "forward": lambda: eng.forward_linear(None, fetch=False)}
        k2 = max(3, min(args.steps, 10))
        for _ in range(2):
            for nm in st2: calls[nm]()
        torch.cuda.synchronize()
        ev2 = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in st2] for _ in range(k2)]
        t1 = time.perf_counter()
        for s_ in range(k2):
            for i, nm in enumerate(st2):
                ev2[s_][i][0].record(stream); calls[nm](); ev2[s_][i][1].record(stream)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        ms2 = {nm: float(np.mean([ev2[s_][i][0].elapsed_time(ev2[s_][i][1]) for s_ in range(k2)])) for i, nm in enumerate(st2)}
        secondary = {"value": B * k2 / el2, "unit": "trajectory-iterations/s", "steps": k2, "ms_per_step": 1e3 * el2 / k2,
                     "kernels": {"backward": eng.backward_variant, "forward": eng.forward_variant}, "stage_ms": ms2,
                     "stage_algorithmic_GBps": {k: ab[k] * B / (ms2[k] * 1e-3) / 1e9 for k in st2},
                     "backward_roofline_frac": ab["backward"] * B / (ms2["backward"] * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        total_traj = B * world
        value = total_traj * args.steps / elapsed
        dom = "backward"
        # HBM bytes of the dominant kernel from the PMC passes (profiles/r01_pmc_traffic.json: separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction) -- same workload only
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            wl = pmc["workload"]
            if wl["task"] == args.task and wl["T"] == T and wl["batch"] == B and not args.generic:
                traffic = pmc["kernels"]["backward_fused" if fused else dom]["traffic_bytes"]
        except Exception:
            traffic = None
        kb = ab_fused if fused else ab          # bytes each launched kernel must move
        # roofline numerator: SURVEY.md 8(d)'s riccati_bwd figure (A,B,l_* in, K,k out) in both modes, so the
        # fraction stays comparable; the fused kernel's own compulsory HBM I/O is reported beside it
        achieved = ab[dom] * B / (stage_ms[dom] * 1e-3) / 1e9
        out = {
            "metric": "iLQR iterations/sec (Panda 7-DoF, T=3000)" if args.task == "panda_reaching" and T == 3000
                      else f"iLQR iterations/sec ({args.task}, T={T})",
            "value": value, "unit": "trajectory-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.task} T={T} set_interval({args.min_N}) keypoints={Kp} "
                                   f"batch={B}/GPU ({uniq} distinct seeds tiled), 6 alphas, lambda={p['lam']}"
                                   + (", fused sweeps (a4+a6 inside a7/a8)" if fused else ""),
                       "batch_per_gpu": B, "global_batch": total_traj, "horizon": T,
                       "kernels": variants,
                       "valid_backward_passes": n_ok, "parallelism": f"traj-shard x{world}"},
            "batch_iterations_per_s": args.steps / elapsed,
            "stage_ms": stage_ms,
            "stage_algorithmic_GBps": {k: kb[k] * B / (stage_ms[k] * 1e-3) / 1e9 for k in stages},
            "pipeline_algorithmic_GBps": sum(ab.values()) * B / (elapsed / args.steps) / 1e9,
            "roofline": {"bound": "hbm", "kernel": f"backward ({variants['backward']})", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab[dom] * B, "avg_launch_ms": stage_ms[dom],
                         "kernel_compulsory_bytes_per_launch": kb[dom] * B,
                         "iteration_algorithmic_GBps": sum(ab.values()) * B / (elapsed / args.steps) / 1e9,
                         "iteration_frac_of_hbm_peak": sum(ab.values()) * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        # second ceiling of the dominant kernel: the SIMD's FP64 unit.  Issued v_mfma_f64_16x16x4 per trajectory-step
        # from the PMC pass (profiles/r01_pmc_counters.json), 2048 flop and 64 busy cycles each = 32 flop/clk/SIMD
        # (profiles/r01_fp64_mfma_probe.txt) -> chip peak 1024 SIMDs x 32 x 2.4 GHz.  Issued, not algorithmic, flops:
        # the control-side products fill 7 of a tile's 16 columns.
        try:
            if fused and args.task == "panda_reaching" and not args.generic:
                cnt = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_counters.json")))["derived"]["backward_fused"]
                issued = cnt["mfma_per_step_per_trajectory"] * 2048.0 * T * B
                peak_tf = 1024 * 32 * 2.4e9 / 1e12
                ach_tf = issued / (stage_ms[dom] * 1e-3) / 1e12
                out["fp64_unit"] = {"kernel": out["roofline"]["kernel"], "issued_mfma_TFLOPs": ach_tf, "peak_TFLOPs": peak_tf,
                                    "frac": ach_tf / peak_tf,
                                    "mfma_per_trajectory_step": cnt["mfma_per_step_per_trajectory"]}
        except Exception:
            pass
        if secondary is not None:
            out["materialising_pipeline"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task, T, args.min_N)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
                # a single trajectory cannot use 16 CPU threads either: the latency comparison for --batch 1
                if out["cpu_baseline"].get("single_thread_value"):
                    out["gpu_over_cpu_single_thread"] = value / out["cpu_baseline"]["single_thread_value"]
            except Exception as ex:   # the baseline is reporting only; never hide the GPU number
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
