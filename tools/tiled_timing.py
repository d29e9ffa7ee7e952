"""configs[2] and configs[4] on the tiled kernels (GPU box): python tools/tiled_timing.py"""
import sys, time, json, subprocess
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
import numpy as np
import torch
import bench
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for key, (kind, task, Ts, Bs, ks) in {"configs[2] pushing n=20 B=64": ("adaptive_jerk", "panda_pushing", 3000, 64, 5),
                                      "light_clutter n=38 B=64": ("set_interval", "light_clutter_push", 2000, 64, 3),
                                      "configs[4] n=62 B=128": ("iterative_error", "high_dof_push", 5000, 128, 3)}.items():
    ps, ps0, ds = bench.build_problem(kind, Bs, Ts, 5, task, distinct=False)
    rs = bench.time_config(torch, stream, 0, ps, ks, 1, True, False)
    pc = bench.parity_check(ps0, rs["eng"], 1, tiled=True)
    rs["eng"].close()
    print(key, rs["variants"]["backward"], {k: round(v, 3) for k, v in rs["stage_ms"].items()}, "K err %.1e" % pc["max_rel_err_K"], flush=True)
