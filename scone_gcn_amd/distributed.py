"""Data-parallel sharding of a trajectory batch over the GPUs of one node (one process per GPU).

The reference is single-process; its batch is a boolean mask over all N trajectories (STM:313-322) and the
loss is a mean over the masked ones (STM:54).  Trajectories are independent, so the batch shards with NO
data-path collective: every rank holds the full operators and weights, processes its slice of the masked
indices, and the only exchange is ONE all-reduce (sum) of the flat weight-gradient buffer per optimiser
step -- RCCL over xGMI when the backend is "nccl".  The ridge term and the Adam update are applied identically
on every rank after the reduction, so the replicas stay bit-identical.
"""
import numpy as np
import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_indices(idx, rank, world_size):
    """Contiguous, balanced split of the (sorted) masked indices; independent of how many GPUs there are in the
    sense that the union over ranks is always exactly `idx`."""
    idx = np.asarray(idx)
    bounds = np.linspace(0, len(idx), world_size + 1).astype(np.int64)
    return idx[bounds[rank]:bounds[rank + 1]]


def all_reduce_sum_(flat, group=None, force=False):
    """In-place sum of the flat gradient buffer over ranks (no-op for a single process, unless `force`: a one-rank
    process group still runs the backend's collective -- the RCCL smoke test uses that on a one-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal on a box without RCCL peers: gloo reduces a host copy (tiny: the flat gradient buffer)
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def data_parallel_grad(idx, grad_fn, flat_grad, group=None, force=False):
    """Generic DP step: `grad_fn(local_idx, total_count)` must ACCUMULATE into `flat_grad` the gradient of
    -sum_{n in local_idx} <logp_n, y_n> / total_count; afterwards flat_grad holds the full-batch gradient of the
    data term on every rank.  Returns the local index slice (for callers that also want the local loss)."""
    rank, ws = world(group)
    local = shard_indices(idx, rank, ws)
    flat_grad.zero_()
    if len(local):
        grad_fn(local, len(idx))
    all_reduce_sum_(flat_grad, group, force)
    return local


def all_reduce_scalar(value, device, group=None):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    all_reduce_sum_(t, group)
    return float(t.item())


def all_max(value, device=None, group=None, force=False):
    """Maximum of a host scalar over ranks (bench.py: the step time every rank reports is the slowest rank's).  The
    reduction runs on `device` (a CUDA device under RCCL, the host under gloo)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if device is None:
            device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([float(value)], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t.item())
    return float(value)
