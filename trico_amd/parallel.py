"""Multi-GPU plumbing: one process per GPU, independent meshes/streams per rank (the format has no
intra-stream parallelism to exchange), and ONE exchange step: gathering the finished .trc archives on
a root rank over RCCL/xGMI.  torch.distributed is used as the transport only."""
import torch

from . import api


def wrap_device_bytes(ptr, nbytes, device):
    """uint8 CUDA tensor aliasing `nbytes` at device address `ptr` (no copy); the owner keeps it alive."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def gather_archives(dist, archive_ptr, archive_size, device, dst=0):
    """All ranks call this with their device-resident archive.  Exchange: all_gather of the sizes (one
    int64 per rank), then every non-root rank sends exactly its bytes to `dst` and the root receives each
    archive at its final offset of one contiguous buffer (point-to-point over xGMI, no padding).
    Returns on dst: (tensor with all archives back to back, list of sizes); elsewhere: None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([archive_size], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes_h = sizes.tolist()
    local = wrap_device_bytes(archive_ptr, archive_size, device)
    if rank == dst:
        out = torch.empty(int(sum(sizes_h)), dtype=torch.uint8, device=device)
        offs = [0]
        for s in sizes_h:
            offs.append(offs[-1] + int(s))
        out[offs[rank]:offs[rank + 1]].copy_(local)
        reqs = []
        for r in range(world):
            if r != dst:
                reqs.append(dist.irecv(out[offs[r]:offs[r + 1]], src=r))
        for q in reqs:
            q.wait()
        return out, sizes_h
    dist.send(local, dst=dst)
    return None
