"""The RCCL leg of the multi-GPU path as far as one GPU can exercise it: process group over the nccl backend with
world size 1, the device-pointer wrapper around a device-resident archive, all_gather of sizes and the root's
assembly in trico_amd.parallel.gather_archives.  (The point-to-point part is covered on gloo, world 2 and 3.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import torch.distributed as dist
from trico_amd import api, meshgen
from trico_amd.parallel import gather_archives, wrap_device_bytes, split_archives
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
v, t = meshgen.grid(64, 32)
a = api.Archive.open_for_writing(1 << 16, device=True)
assert a.write("vertices", torch.from_numpy(v).cuda(), 64 * 32) == 1
assert a.write("triangles", torch.from_numpy(t.view(np.int32)).cuda(), 2 * 64 * 32) == 1
size = a.get_size()
w = wrap_device_bytes(a.get_buffer_pointer(), size, torch.device("cuda", 0))
assert w.is_cuda and w.numel() == size and w.data_ptr() == a.get_buffer_pointer()
out, sizes = gather_archives(dist, w, dst=0)
dist.barrier()
torch.cuda.synchronize()
assert sizes == [size]
parts = split_archives(out, sizes)
assert bytes(parts[0].cpu().numpy().tobytes()) == a.tobytes()
a.close()
dist.destroy_process_group()
print("OK", size)
'''


def test_nccl_world1_gather(tmp_path):
    import socket
    with socket.socket() as sk:                 # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", WORKER % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


SHARD_WORKER = r'''
import os, sys, hashlib, json
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import torch.distributed as dist
from trico_amd import api
from trico_amd.parallel import sharded_write, hip_unit_encoder, CComm
from streams import mesh_streams
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
hashes = json.load(open(os.path.join(%r, "tests", "golden", "hashes.json")))
enc = hip_unit_encoder(api)
comm = CComm(api, 0, 1, lambda b: b)          # RCCL through the C-ABI (world 1: the id needs no broadcast)
for kind in ("grid", "multi"):
    streams = mesh_streams(kind, 1000, 1000)
    dev = [(n, torch.from_numpy(a.view(np.uint8)).cuda(), c) for n, a, c in streams]
    for transport in ("torch", "c"):
        a = sharded_write(dist, api, dev, enc, root=0, device_archive=True, gather=(lambda t: comm.gather(t, 0)) if transport == "c" else None)
        blob = a.tobytes()
        a.close()
        g = hashes["%%s_1000x1000" %% kind]
        assert len(blob) == g["size"], (kind, transport, len(blob), g["size"])
        assert hashlib.sha256(blob).hexdigest() == g["sha256"], (kind, transport)
comm.close()
enc.close()
dist.destroy_process_group()
print("OK")
'''


def test_stream_sharded_write_on_the_gpu_matches_the_reference_archive():
    """The unit encoders (trico_hip_fpc_encode_component / _int_encode_plane), the assembly
    (trico_hip_append_encoded_stream) and both exchanges (torch.distributed nccl, and RCCL behind trico_hip_comm_*) at
    world size 1: the archive assembled from separately encoded units is the reference's (sha256 of config 1 and its
    multi sibling)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", SHARD_WORKER % (ROOT, ROOT, ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-500:] + r.stderr[-2500:]


GLOO_SHARD_WORKER = r'''
import os, sys, hashlib, json
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import torch.distributed as dist
from trico_amd import api
from trico_amd.parallel import sharded_write, hip_unit_encoder
from streams import mesh_streams
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
hashes = json.load(open(os.path.join(%r, "tests", "golden", "hashes.json")))
enc = hip_unit_encoder(api)
for kind in ("grid", "multi"):
    streams = mesh_streams(kind, 1000, 1000)
    dev = [(n, torch.from_numpy(a.view(np.uint8)).cuda(), c) for n, a, c in streams]
    a = sharded_write(dist, api, dev, enc, root=0, device_archive=False)
    if rank == 0:
        blob = a.tobytes()
        a.close()
        g = hashes["%%s_1000x1000" %% kind]
        assert len(blob) == g["size"] and hashlib.sha256(blob).hexdigest() == g["sha256"], kind
    else:
        assert a is None
enc.close()
dist.barrier()
dist.destroy_process_group()
print("RANK", rank, "OK")
'''


@pytest.mark.parametrize("world", [2, 4])
def test_stream_sharded_write_with_several_ranks_on_one_gpu(world):
    """More than one rank with the REAL unit encoders: `world` processes share this box's GPU (every rank encodes its units of the
    grid's 7 and the multi mesh's 16 on it), the exchange runs over gloo through host memory, rank 0 assembles.  The archive is the
    reference's (sha256 of config 1 and its multi sibling).  What this does not exercise is RCCL between GPUs: there is one GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, "-c", GLOO_SHARD_WORKER % (ROOT, ROOT, ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "OK" in o, "rank %d: %s %s" % (r, o[-300:], e[-1500:])
