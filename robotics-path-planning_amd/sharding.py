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


def gather_results(dist, path_cost, n_nodes, status, device=None, device_table=None):
    """all_gather of the result table; returns (path_cost, n_nodes, status) over all ranks, rank-major order.
    With dist=None (single process) returns the inputs.  `device_table` = a planned `_abi.Handle`: the packed
    16-byte records {f64 path_cost, i32 n_nodes, i32 status} are then copied device -> device into the tensor the
    collective sends (rrtx_copy_results_device), with no round trip through host memory; otherwise (CPU / gloo tests)
    the host arrays are packed into the same record layout."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(path_cost), np.asarray(n_nodes), np.asarray(status)
    import torch
    n = len(path_cost)
    if device_table is not None and device is not None:
        t = torch.empty((n, 2), dtype=torch.int64, device=device)
        device_table.copy_results_device(t.data_ptr(), t.numel() * 8)
    else:
        t = torch.from_numpy(pack_records(path_cost, n_nodes, status))
        if device is not None:
            t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return unpack_records(torch.cat(out).cpu().numpy())


RECORD = np.dtype([("path_cost", "<f8"), ("n_nodes", "<i4"), ("status", "<i4")])   # rppk::Result, 16 bytes


def pack_records(path_cost, n_nodes, status):
    """(n, 2) int64 view of the 16-byte result records."""
    rec = np.zeros(len(path_cost), dtype=RECORD)
    rec["path_cost"], rec["n_nodes"], rec["status"] = path_cost, n_nodes, status
    return rec.view(np.int64).reshape(-1, 2)


def unpack_records(a):
    rec = np.ascontiguousarray(a, dtype=np.int64).reshape(-1).view(RECORD)
    st = rec["status"].astype(np.int64)
    pc = np.where((st & 2) != 0, rec["path_cost"], np.inf)   # rrtx_get_results: no path <=> +inf
    return pc, rec["n_nodes"].astype(np.int64), st


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


def gather_floats(dist, values, device=None):
    """all_gather of a short float vector per rank; returns [[values of rank 0], [values of rank 1], ...]."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [[float(v) for v in values]]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.cpu().tolist()] for o in out]


def split_contiguous(n, shards):
    """[lo, hi) of each of `shards` contiguous blocks of n instances, sizes differing by at most one (the first n %
    shards blocks are the longer ones); instance i of the whole batch is instance i - lo of its shard."""
    base, extra = divmod(int(n), int(shards))
    out, lo = [], 0
    for r in range(int(shards)):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out
