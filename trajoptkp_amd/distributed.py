"""Multi-GPU layer: trajectories (MPC replans / initial conditions) are independent units, so the
batch is sharded across ranks -- one process per GPU, one Engine per process -- with NO data-path
collective.  The only exchange is the line-search cost reduction named by BASELINE.json's
north_star: one all-reduce of 8 doubles per iteration,
    [sum_b J_pred(alpha_1..6), sum_b delta_J, number of trajectories with a valid backward pass],
over torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
The reference itself selects alpha per trajectory (src/Optimiser/iLQR.cpp:490-502); the reduced
vector is the batch-level monitor (expected reduction, best common alpha, failure count).
"""
import numpy as np


def shard_range(n_traj, rank, world):
    """Contiguous block of trajectories owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_traj, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_linesearch(cost_pred, delta_J, status):
    """Local 8-vector from per-trajectory results (torch tensors on any device, or numpy)."""
    import torch
    cp = torch.as_tensor(cost_pred)
    out = torch.zeros(8, dtype=torch.float64, device=cp.device)
    ok = (torch.as_tensor(status).to(cp.device) == 0)
    n_alpha = cp.shape[1]
    if n_alpha > 6:
        raise ValueError("the reduction vector carries at most 6 alphas")
    # torch.where, not multiplication: a trajectory whose backward pass failed keeps stale gains and its rollout may
    # hold inf/NaN (inf * 0 = NaN would poison the sum); k_pack_linesearch skips such rows the same way
    zero = torch.zeros((), dtype=cp.dtype, device=cp.device)
    out[:n_alpha] = torch.where(ok[:, None], cp, zero).sum(0)
    dJ = torch.as_tensor(delta_J).to(cp.device)
    out[6] = torch.where(ok, dJ, zero.to(dJ.dtype)).sum()
    out[7] = ok.sum()
    return out


def allreduce_linesearch(vec8, group=None):
    """In-place SUM all-reduce of the 8-vector; no-op when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec8, op=dist.ReduceOp.SUM, group=group)
    return vec8


def best_common_alpha(vec8, alphas):
    """Batch-level line-search summary: alpha index with the lowest summed predicted cost change."""
    v = np.asarray(vec8.cpu() if hasattr(vec8, "cpu") else vec8, dtype=np.float64)
    i = int(np.argmin(v[:len(alphas)]))
    return i, float(alphas[i]), float(v[i]), int(v[7])
