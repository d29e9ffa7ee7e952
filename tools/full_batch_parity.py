"""Every trajectory of the headline workload against the CPU oracle (GPU box): python tools/full_batch_parity.py [B] [T]

bench.py's parity_check looks at the first 8 of the 1024 distinct seeds; this looks at all of them.  The oracle results are
computed first, by forked workers, one seed at a time (synth.make_problem(first_b=b) is the same trajectory as row b of the
batch), into shared arrays; only then is the GPU touched."""
import os, sys, time
import multiprocessing as mp
import numpy as np
sys.path.insert(0, ".")
import bench
from trajoptkp_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
TASK, MIN_N = "panda_reaching", 5
probe = synth.make_problem(task=TASK, T=T, batch=1, min_N=MIN_N)
n, m = probe["n"], probe["m"]
oK = bench._shared((B, T, n * m)); ok = bench._shared((B, T, m)); oC = bench._shared((B, 6)); oJ = bench._shared((B,))
oS = bench._shared((B,), np.int32)


def work(b):
    from oracle import pipeline
    q = synth.make_problem(task=TASK, T=T, batch=1, min_N=MIN_N, first_b=b)
    o = pipeline.run_trajectory(q, 0)
    oK[b] = o["K"].reshape(T, -1); ok[b] = o["k"].reshape(T, -1); oC[b] = o["cost_pred"]; oJ[b] = o["delta_J"]; oS[b] = o["status"]
    return b


t0 = time.time()
p = bench.distinct_problem(TASK, T, B, MIN_N, cache=os.environ.get("KPILQR_WORKLOAD_CACHE"))
with mp.get_context("fork").Pool(max(1, min(16, len(os.sched_getaffinity(0))))) as pool:
    for i, _ in enumerate(pool.imap_unordered(work, range(B), chunksize=4)):
        if i % 128 == 0:
            print(f"oracle: {i} of {B} trajectories ({time.time() - t0:.0f} s)", flush=True)
print(f"workload + oracle: {time.time() - t0:.1f} s", flush=True)

import torch
from trajoptkp_amd import Engine
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
rel = lambda a, r: float(np.max(np.abs(a - r)) / max(float(np.max(np.abs(r))), 1e-300))
for label, env in (("one wave per trajectory, payload differenced inside the backward sweep", {}),
                   ("column store differenced by k_fd_kp_difference first", {"KPILQR_FUSED_RAW": "0"})):
    os.environ.pop("KPILQR_FUSED_RAW", None)
    os.environ.update(env)
    with Engine(p["dof"], p["m"], T, p["nr"], batch=B, device=0, stream=stream.cuda_stream, fused=True) as eng:
        synth.upload(eng, p, kp_ordered=True)
        st, dJ = eng.backward(np.full(B, p["lam"]), 100)
        K, k = eng.gains()
        cost = eng.forward_linear(np.array([(i / 6.0) ** 2 for i in range(1, 7)]))
        var = (eng.backward_variant, eng.forward_variant)
    eK = np.array([rel(K[b].reshape(T, -1), oK[b]) for b in range(B)])
    ek = np.array([rel(k[b].reshape(T, -1), ok[b]) for b in range(B)])
    eC = np.array([rel(cost[b], oC[b]) for b in range(B)])
    eJ = np.abs(dJ - oJ) / np.abs(oJ)
    print(f"{label}: {var[0]} / {var[1]}, B={B}, T={T}, {B} distinct seeds")
    print(f"  worst relative error over ALL trajectories: K {eK.max():.2e} (trajectory {int(eK.argmax())}), k {ek.max():.2e}, "
          f"predicted costs {eC.max():.2e}, delta_J {eJ.max():.2e}; median K {np.median(eK):.2e}; "
          f"status mismatches {int(np.count_nonzero(np.asarray(st) != oS))}", flush=True)
    assert eK.max() < 1e-9 and ek.max() < 1e-9 and eC.max() < 1e-9 and eJ.max() < 1e-9 and np.array_equal(np.asarray(st), oS)
print("all trajectories within 1e-9 of the oracle")
