"""Instance sharding across GPUs: planning instances are independent, so rank r of W simply owns
instances [r*B, (r+1)*B) -- seeds r*B+1 ... -- and nothing is exchanged while planning.  The only collective is the
gather of the per-instance result records {path_cost f64, n_nodes, status} (16 B each) at the end, plus the
max/sum reductions the benchmark needs.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
import numpy as np


def shard_seeds(rank, per_rank, base=1):
    """Seeds (== `random.seed(s)` values) of the instances owned by `rank` (weak scaling: per_rank each)."""
    return [base + rank * per_rank + i for i in range(per_rank)]


def instance_owner(instance, per_rank):
    return instance // per_rank


def gather_results(dist, path_cost, n_nodes, status, device=None):
    """all_gather of the result table; returns (path_cost, n_nodes, status) over all ranks, rank-major order.
    With dist=None (single process) returns the inputs."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(path_cost), np.asarray(n_nodes), np.asarray(status)
    import torch
    rec = np.stack([np.asarray(path_cost, dtype=np.float64), np.asarray(n_nodes, dtype=np.float64),
                    np.asarray(status, dtype=np.float64)], axis=1)
    t = torch.from_numpy(rec)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    allr = torch.cat(out).cpu().numpy()
    return allr[:, 0], allr[:, 1].astype(np.int64), allr[:, 2].astype(np.int64)


def reduce_max(dist, value, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_int(dist, values, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(v) for v in values]
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def all_agree_min(dist, flag, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(flag)
    import torch
    t = torch.tensor([int(flag)], dtype=torch.int32)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())
