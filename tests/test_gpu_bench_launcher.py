"""`python bench.py --gpus N` without a launcher (the driver's contract): the script starts its N ranks itself, as a
child process started before anything touches the GPU, and refuses a launcher that started another number of ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--global-batch", "64", "--T", "200", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"]


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KPILQR_DIST_BACKEND")
           and not k.startswith("TORCHELASTIC") and not k.startswith("KPILQR_FUSED")}
    env.update(extra)
    return env


def test_bench_gpus_2_starts_two_ranks_itself_and_shards_the_global_batch():
    import torch
    ndev = torch.cuda.device_count()                 # (does not initialise HIP)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                       # rank 0's line, once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 64 and out["config"]["batch_per_gpu"] == 32          # 32 trajectories on rank 0
    assert out["config"]["valid_backward_passes_rank0"] == 32 and out["parity_check"]["pass"]
    assert out["value"] > 0 and abs(out["value"] - 64 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-4 * out["value"]    # (the line carries six digits)
    if ndev >= 2:
        assert out["rccl_ranks"] == 2 and "RCCL" in out["collective"], out["collective"]
    else:   # one GPU: the ranks share it, RCCL refuses such a communicator, the rehearsal's collective runs over gloo and says so
        assert out["rccl_ranks"] == 0 and "REHEARSAL" in out["collective"], out["collective"]


def test_bench_refuses_a_launcher_with_another_rank_count():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, cwd=ROOT,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "--gpus 2 but the launcher started 1 rank" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
