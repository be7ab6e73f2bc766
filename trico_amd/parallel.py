"""Multi-GPU plumbing: one process per GPU, independent meshes/streams per rank (the .trc format has no
intra-stream parallelism to exchange), and ONE exchange step: gathering the finished archives on a
root rank over RCCL/xGMI.  torch.distributed is used as the transport only; the same
code runs on gloo/CPU tensors (tests/test_dist_gloo.py)."""
import torch


def wrap_device_bytes(ptr, nbytes, device):
    """uint8 tensor aliasing `nbytes` at address `ptr` of `device` (no copy); the owner keeps it alive."""
    if torch.device(device).type == "cpu":
        import ctypes
        return torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(int(ptr)), dtype=torch.uint8)
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


# ---- one mesh over several GPUs: stream-sharded encoding ---------------------------------------------------------------
# The independent units of a mesh are the components of its real streams and the byte planes of its integer streams
# (trico.c:229-260, 346-368 compress them one after the other).  Rank r encodes units r, r + world, ...; one exchange
# collects the payloads on the root, which frames them into ONE archive with trico_hip_append_encoded_stream - byte for
# byte the archive a single trico_write_* sequence produces.

# name -> (stream tag, arity or None, element width, count multiplier of the stored count field, elements per count)
STREAM_SHAPES = {
    "vertices": (1, 3, 4, 1, 1), "vertices_double": (2, 3, 8, 1, 1),
    "triangles": (3, None, 4, 1, 3), "triangles_long": (4, None, 8, 1, 3),
    "uv_per_vertex": (5, 2, 4, 1, 1), "uv_per_triangle": (7, 2, 4, 3, 3),
    "vertex_normals": (9, 3, 4, 1, 1), "vertex_normals_double": (10, 3, 8, 1, 1),
    "triangle_normals": (11, 3, 4, 1, 1), "triangle_normals_double": (12, 3, 8, 1, 1),
    "vertex_colors": (13, None, 4, 1, 1), "triangle_colors": (14, None, 4, 1, 1),
    "attributes_float": (15, 1, 4, 1, 1), "attributes_double": (16, 1, 8, 1, 1),
    "attributes_uint8": (17, None, 1, 1, 1), "attributes_uint16": (18, None, 2, 1, 1),
    "attributes_uint32": (19, None, 4, 1, 1), "attributes_uint64": (20, None, 8, 1, 1),
}


def stream_units(streams):
    """[(stream index, unit index)] of a list of (name, data, count) in archive order."""
    units = []
    for si, (name, _, _) in enumerate(streams):
        _, arity, width, _, _ = STREAM_SHAPES[name]
        units += [(si, u) for u in range(arity if arity is not None else width)]
    return units


def hip_unit_encoder(api):
    """Encodes one unit on this rank's GPU through the C-ABI; returns a uint8 CUDA tensor holding the payload."""
    import ctypes
    L = api.lib()
    ctx = L.trico_hip_ctx_create()
    if not ctx:
        raise RuntimeError(api.last_error())

    def encode(name, data, count, unit):
        _, arity, width, _, per = STREAM_SHAPES[name]
        size = ctypes.c_uint32(0)
        if arity is not None:
            n = count * (3 if name == "uv_per_triangle" else 1)
            ok = L.trico_hip_fpc_encode_component(ctx, api.ptr(data), n, arity, width, unit, ctypes.byref(size))
        else:
            ok = L.trico_hip_int_encode_plane(ctx, api.ptr(data), count * per, width, unit, ctypes.byref(size))
        if not ok:
            raise RuntimeError(api.last_error())
        out = torch.empty(size.value, dtype=torch.uint8, device="cuda")
        if size.value and not L.trico_hip_fetch_payload(ctx, 0, out.data_ptr()):
            raise RuntimeError(api.last_error())
        return out

    encode.close = lambda: L.trico_hip_ctx_destroy(ctx)
    return encode


def collective_device(dist):
    """Device the tensors of a collective must live on: the current GPU under nccl (= RCCL), the CPU under gloo.  A rank
    that owns no unit (world > units: the 7 units of a vertices + triangles mesh on 8 GPUs) still takes part in every
    collective, and must do so with tensors of the backend's device."""
    backend = str(dist.get_backend()).lower()
    if "nccl" in backend and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def sharded_write(dist, api, streams, encode_unit, root=0, device_archive=False, gather=None):
    """All ranks call this with the same `streams` = [(name, data, count)] (every rank holds, or can read, the whole
    mesh).  Each rank encodes its units with `encode_unit(name, data, count, unit) -> uint8 tensor`, the payloads are
    gathered on `root` (default transport: gather_archives over torch.distributed; `gather` may be given instead, e.g.
    the RCCL entry of the C-ABI), and the root returns the assembled archive (api.Archive); other ranks return None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    units = stream_units(streams)
    mine = shard_units(len(units), world, rank)
    payloads = [encode_unit(streams[units[u][0]][0], streams[units[u][0]][1], streams[units[u][0]][2], units[u][1]) for u in mine]
    dev = collective_device(dist)
    # (gloo moves host tensors: payloads encoded on a GPU travel through host memory then; under nccl they stay where they are)
    payloads = [p if p.device.type == dev.type else p.to(dev) for p in payloads]
    # every rank's unit sizes, padded to the same length: one small all-gather
    per = (len(units) + world - 1) // world
    mysz = torch.zeros(per, dtype=torch.int64, device=dev)
    for k, p in enumerate(payloads):
        mysz[k] = p.numel()
    allsz = [torch.zeros(per, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, mysz)
    local = torch.cat(payloads) if payloads else torch.empty(0, dtype=torch.uint8, device=dev)
    res = (gather or (lambda t: gather_archives(dist, t, dst=root)))(local)
    if rank != root:
        return None
    buf, rank_sizes = res
    # locate unit u inside the gathered buffer: rank u % world, its (u // world)-th payload
    rank_off, o = [], 0
    for s in rank_sizes:
        rank_off.append(o)
        o += s
    usz = [int(allsz[u % world][u // world].item()) for u in range(len(units))]
    uoff = []
    for u in range(len(units)):
        r, k = u % world, u // world
        uoff.append(rank_off[r] + sum(int(allsz[r][j].item()) for j in range(k)))
    total = 8 + sum(5 + 4 * (STREAM_SHAPES[n][1] or STREAM_SHAPES[n][2]) for n, _, _ in streams) + sum(usz)
    a = api.Archive.open_for_writing(total + 64, device=device_archive)
    u = 0
    for name, _, count in streams:
        tag, arity, width, mult, _ = STREAM_SHAPES[name]
        k = arity if arity is not None else width
        parts = [buf[uoff[u + j]: uoff[u + j] + usz[u + j]] for j in range(k)]
        ok = a.append_encoded_stream(tag, count * mult, parts, usz[u: u + k])
        if ok != 1:
            raise RuntimeError("append_encoded_stream(%s): %s" % (name, api.last_error()))
        u += k
    return a


class CComm:
    """RCCL communicator behind the C-ABI (trico_hip_comm_*): the exchange step without torch.distributed in the data
    path.  The 128-byte id is created on rank 0 and handed to the others with `share(bytes) -> bytes` (any broadcast)."""

    def __init__(self, api, rank, world, share):
        import ctypes
        self.api, self.rank, self.world = api, rank, world
        L = api.lib()
        idbuf = (ctypes.c_uint8 * 128)()
        if rank == 0 and not L.trico_hip_comm_unique_id(idbuf):
            raise RuntimeError(api.last_error())
        ident = share(bytes(idbuf))
        idbuf = (ctypes.c_uint8 * 128).from_buffer_copy(ident)
        self.h = L.trico_hip_comm_create(idbuf, rank, world)
        if not self.h:
            raise RuntimeError(api.last_error())

    def gather(self, local, root=0):
        """local: uint8 CUDA tensor.  Root: (tensor with every rank's bytes back to back, sizes); others: None."""
        import ctypes
        L = self.api.lib()
        sizes = (ctypes.c_uint64 * self.world)()
        # the root cannot size its buffer before the exchange: first call with capacity 0 learns the sizes (nothing moves),
        # second call moves the bytes
        L.trico_hip_comm_gather(self.h, local.data_ptr() if local.numel() else None, local.numel(), root, None, 0, sizes)
        total = sum(sizes)
        out = torch.empty(total, dtype=torch.uint8, device=local.device) if self.rank == root else None
        ok = L.trico_hip_comm_gather(self.h, local.data_ptr() if local.numel() else None, local.numel(), root,
                                     out.data_ptr() if out is not None and total else None, total if self.rank == root else 0, sizes)
        if not ok and total:
            raise RuntimeError(self.api.last_error())
        return (out, [int(s) for s in sizes]) if self.rank == root else None

    def close(self):
        if self.h:
            self.api.lib().trico_hip_comm_destroy(self.h)
            self.h = None
