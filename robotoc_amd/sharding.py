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


class RcclComm:
    """An ncclComm_t of this process's own (librccl through ctypes) for the C ABI's rtoc_gather_directions -- what a C++
    host holds; torch.distributed keeps its communicator to itself.  Rank 0 draws the ncclUniqueId and the ranks exchange
    it through the torch.distributed group that is already up.  One process per GPU, the device set beforehand."""

    def __init__(self, world, rank):
        import ctypes as C
        import os

        class _UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]   # rccl.h: NCCL_UNIQUE_ID_BYTES

        lib = None
        for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
            try:
                lib = C.CDLL(name, mode=os.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError("librccl not found")
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = _UniqueId()
        if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        box = [bytes(bytearray(uid))]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
            C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)
        self._lib, self.handle = lib, comm.value

    def close(self):
        if self.handle:
            self._lib.ncclCommDestroy(self.handle)
            self.handle = None
