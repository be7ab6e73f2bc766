"""Multi-GPU plumbing: one process per GPU, independent meshes/streams per rank (the .trc format has no
intra-stream parallelism to exchange), and ONE exchange step: gathering the finished archives on a
root rank over RCCL/xGMI.  torch.distributed is used as the transport only; the same
code runs on gloo/CPU tensors (tests/test_dist_gloo.py)."""
import torch


def wrap_device_bytes(ptr, nbytes, device):
    """uint8 CUDA tensor aliasing `nbytes` at device address `ptr` (no copy); the owner keeps it alive."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def shard_units(n_units, world, rank):
    """Independent units (meshes / archives / component streams) owned by `rank`: round-robin."""
    return list(range(rank, n_units, world))


def gather_archives(dist, local, dst=0):
    """All ranks call this with `local`, a uint8 tensor holding their archive bytes (CUDA for RCCL, CPU for
    gloo).  Exchange: all_gather of the sizes (one int64 per rank), then every non-root rank sends exactly its
    bytes to `dst`, which receives each archive at its final offset of one contiguous buffer (point-to-point,
    no padding: over xGMI every sender has its own link to the root).
    Returns on dst: (tensor with all archives back to back, list of sizes); elsewhere: None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = local.device
    mine = torch.tensor([local.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes_h = [int(s.item()) for s in sizes]
    if rank == dst:
        out = torch.empty(sum(sizes_h), dtype=torch.uint8, device=device)
        offs = [0]
        for s in sizes_h:
            offs.append(offs[-1] + s)
        out[offs[rank]:offs[rank + 1]].copy_(local)
        reqs = [dist.irecv(out[offs[r]:offs[r + 1]], src=r) for r in range(world) if r != dst and sizes_h[r]]
        for q in reqs:
            q.wait()
        return out, sizes_h
    if local.numel():
        dist.send(local, dst=dst)
    return None


def split_archives(buf, sizes):
    """Inverse bookkeeping on the root: views of the individual archives inside the gathered buffer."""
    out, o = [], 0
    for s in sizes:
        out.append(buf[o:o + s])
        o += s
    return out
