"""configs[2] (pushing n=20, B=64) and walker on the two-tile kernels, for A/B runs of library builds (GPU box, KPILQR_LIB=...)"""
import sys
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
import torch
import bench
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for key, (kind, task, Ts, Bs, ks) in {"pushing n=20 B=64": ("adaptive_jerk", "panda_pushing", 3000, 64, 10),
                                      "walker n=18 B=64": ("set_interval", "walker", 3000, 64, 10)}.items():
    ps, ps0, ds = bench.build_problem(kind, Bs, Ts, 5, task, distinct=False)
    rs = bench.time_config(torch, stream, 0, ps, ks, 2, True, False)
    pc = bench.parity_check(ps0, rs["eng"], 1, tiled=True)
    rs["eng"].close()
    print(key, rs["variants"], {k: round(v, 3) for k, v in rs["stage_ms"].items()}, "K err %.1e cost err %.1e" % (pc["max_rel_err_K"], pc["max_rel_err_cost_pred"]), flush=True)
