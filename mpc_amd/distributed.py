"""Multi-GPU use of the batched planner: one process per GPU, no solve-time exchange.

The instances of a batch are independent, so a node-level batch is split into contiguous
blocks, one per rank; every rank creates its own planner handle on its own device and solves
its block.  The only communication is the optional epilogue below (a few bytes per instance):
gathering {cost, status} on every rank, or picking the instance with the lowest cost over the
whole job.  With the "nccl" backend of torch.distributed (RCCL on ROCm) the tensors stay on
the GPU and travel over xGMI; with "gloo" they are CPU tensors (used by the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """[first, last) of the contiguous block of `total` instances owned by `rank`
    (instance i belongs to rank floor(i * world / total))."""
    first = (rank * total + world - 1) // world
    last = ((rank + 1) * total + world - 1) // world
    return first, last


def _comm_device(t):
    return t.device if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_cost_status(cost, status):
    """All ranks receive the job-wide cost (float64) and status (int32) vectors, in instance
    order.  Blocks may differ in length by one (uneven split): they are padded for the
    collective and trimmed afterwards."""
    world = dist.get_world_size()
    dev = _comm_device(cost)
    n = torch.tensor([cost.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    # one packed buffer per rank: cost bits and status side by side (12 bytes per instance)
    buf = torch.zeros(width, 2, dtype=torch.float64, device=dev)
    buf[:cost.numel(), 0] = cost.to(dev)
    buf[:cost.numel(), 1] = status.to(dev, dtype=torch.float64)
    out = torch.empty(world * width, 2, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, width, 2)
    costs = torch.cat([out[r, :sizes[r], 0] for r in range(world)])
    stats = torch.cat([out[r, :sizes[r], 1] for r in range(world)]).to(torch.int32)
    return costs, stats


def best_cost(cost, status, first):
    """(cost, global instance index) of the solved instance (status 0) with the lowest cost
    over all ranks; ties go to the lower index.  `first` is the global index of this rank's
    first instance.  Returns (inf, -1) when no instance was solved anywhere.

    Two MIN all-reduces: the cost, then the index among the ranks that hold that cost."""
    dev = _comm_device(cost)
    ok = status.to(dev) == 0
    c = torch.where(ok, cost.to(dev), torch.full_like(cost.to(dev), float("inf")))
    if c.numel():
        local_best, local_arg = torch.min(c, dim=0)
    else:
        local_best, local_arg = torch.tensor(float("inf"), dtype=torch.float64, device=dev), torch.tensor(0, device=dev)
    best = local_best.clone().reshape(1)
    dist.all_reduce(best, op=dist.ReduceOp.MIN)
    big = torch.iinfo(torch.int64).max
    mine = bool(c.numel()) and bool((local_best == best[0]).item()) and bool(torch.isfinite(best[0]).item())
    idx = torch.tensor([first + int(local_arg.item()) if mine else big], dtype=torch.int64, device=dev)
    dist.all_reduce(idx, op=dist.ReduceOp.MIN)
    i = int(idx.item())
    return float(best.item()), (i if i != big else -1)
