"""Multi-GPU sharding of independent OCP instances.

Instances have no coupling, so the global instance range is split into contiguous per-rank
shards and every rank runs the same kernels on its shard with NO data-path collective.
The one real exchange step is the gather of the step directions; it is a single
all-gather of the packed direction records (RCCL over xGMI through torch.distributed's
"nccl" backend on GPUs, gloo in the CPU tests; a C++ host uses rtoc_gather_directions).
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous shard [lo, hi) of `total` instances for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_directions(local_dir, total, world, rank):
    """All-gather the per-rank direction records into a [total, stages, stride] tensor.

    local_dir: [n_local, stages, stride] tensor (device or host).  Shards may differ in size by
    one instance, so the exchange pads to the largest shard (one collective, no per-rank loop)."""
    if world == 1:
        return local_dir
    sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
    nmax = max(sizes)
    pad = local_dir
    if local_dir.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local_dir.shape[1:]), dtype=local_dir.dtype,
                          device=local_dir.device)
        pad[:local_dir.shape[0]] = local_dir
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
