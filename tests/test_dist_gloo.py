"""CPU tier: the N>1 path (one process per rank, shard independent units, size exchange + point-to-point
gather of variable-length archives onto rank 0) on the gloo backend with world_size 2, 3 and 8 (the shape of
BASELINE configs[3]: one mesh per rank, seeds GRID_SEED + rank; and 7 stream units over 8 ranks, one rank idle).
The bytes gathered are oracle archives, so the test also shows that per-rank archives are position independent."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, n_units=5):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from trico_amd import meshgen
    from trico_amd.parallel import gather_archives, shard_units, split_archives
    mine = shard_units(n_units, world, rank)
    blobs = []
    for u in mine:
        v, t = meshgen.grid(24 + u, 10, meshgen.GRID_SEED + u)
        a = O.OracleArchive()
        a.write("vertices", v, (24 + u) * 10)
        a.write("triangles", t, 2 * (24 + u) * 10)
        blobs.append(a.tobytes())
        a.close()
    local = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy())
    res = gather_archives(dist, local, dst=0)
    if rank == 0:
        buf, sizes = res
        parts = split_archives(buf, sizes)
        q.put((sizes, [bytes(p.numpy().tobytes()) for p in parts]))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 5), (3, 5), (8, 8), (8, 5)])
def test_gather_archives_gloo(world, n_units, native_libs):
    """(8, 8): BASELINE configs[3]'s shape - eight ranks, one mesh each (seeds GRID_SEED + rank), archives gathered on rank 0;
    (8, 5): three ranks own nothing and still take part in the exchange."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from trico_amd import meshgen
    from trico_amd.parallel import shard_units
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world * 7 + n_units * 3 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_units)) for r in range(world)]
    for p in procs:
        p.start()
    sizes, parts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # expected: per rank, its units' archives concatenated in unit order
    for r in range(world):
        want = b""
        for u in shard_units(n_units, world, r):
            v, t = meshgen.grid(24 + u, 10, meshgen.GRID_SEED + u)
            a = O.OracleArchive()
            a.write("vertices", v, (24 + u) * 10)
            a.write("triangles", t, 2 * (24 + u) * 10)
            want += a.tobytes()
            a.close()
        assert sizes[r] == len(want)
        assert parts[r] == want


# ---- one mesh sharded by stream units over the ranks, assembled into ONE archive on the root --------------------------------
def _oracle_unit_encoder():
    """CPU stand-in for the per-unit HIP encoders (test infrastructure): the oracle's coders on one component / plane."""
    from oracle import oracle as O
    from trico_amd.parallel import STREAM_SHAPES

    def encode(name, data, count, unit):
        _, arity, width, _, per = STREAM_SHAPES[name]
        if arity is not None:
            comp = np.ascontiguousarray(data.reshape(-1, arity)[:, unit])
            pay = O.fpc_encode(comp)
        else:
            plane = np.ascontiguousarray(data.view(np.uint8).reshape(-1, width)[:, unit])
            pay = O.lz4_compress(plane.tobytes())
        return torch.from_numpy(np.frombuffer(bytes(pay), np.uint8).copy())
    return encode


def _streams_for(kind):
    from streams import mesh_streams
    return mesh_streams(kind, 40, 24)


def _shard_worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trico_amd import api
    from trico_amd.parallel import sharded_write
    a = sharded_write(dist, api, _streams_for(kind), _oracle_unit_encoder(), root=0)
    if rank == 0:
        q.put(a.tobytes())
        a.close()
    else:
        assert a is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "grid"), (3, "grid"), (3, "multi"), (8, "grid")])
def test_stream_sharded_archive_is_the_single_rank_archive(world, kind, native_libs):
    """Rank r encodes units r, r + world, ... of ONE mesh (x, y, z, b1..b4; for `multi` 3 + 3 + 2 components and 8 planes);
    the root frames the gathered payloads with trico_hip_append_encoded_stream.  The result must be the archive the
    reference's writer sequence produces (here: the oracle's), byte for byte."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + world * 11 + (os.getpid() % 500) + (7 if kind == "multi" else 0)
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    o = O.OracleArchive()
    for name, data, count in _streams_for(kind):
        o.write(name, data, count)
    want = o.tobytes()
    o.close()
    assert got == want
