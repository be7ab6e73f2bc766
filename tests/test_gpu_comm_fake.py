"""GPU tier: trico_hip_comm_gather (dist.hip) with MORE THAN ONE rank on a one-GPU box.  libtrico_testhooks.so carries a test
transport (trico_hip_comm_create_fake: the ranks are threads of one process, the four RCCL calls the gather makes are played by a
rendezvous in host memory + hipMemcpy between the ranks' device buffers), so that the size exchange, the root's verdict, the offsets
of the point-to-point transfers, empty ranks and a root other than rank 0 are exercised.  The real transport (RCCL) has only ever
run with world size 1 on hardware (tests/test_gpu_dist_single.py): every gpurun box has one GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, sys, threading
import numpy as np
sys.path.insert(0, %(root)r)
import torch
from trico_amd import api
L = api.lib()
assert L.trico_hip_available() == 1
L.trico_hip_comm_create_fake.restype = ctypes.c_void_p
L.trico_hip_comm_create_fake.argtypes = [ctypes.c_int, ctypes.c_int]
W, ROOT_RANK = 4, 2
sizes_want = [1000, 0, 70001, 3]
data = [torch.from_numpy(((np.arange(n, dtype=np.uint32) * (r + 3) + r) %% 251).astype(np.uint8)).cuda() for r, n in enumerate(sizes_want)]
total = sum(sizes_want)
out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
res = {}

def rank(r):
    c = L.trico_hip_comm_create_fake(r, W)
    assert c
    sz = (ctypes.c_uint64 * W)()
    ptr = data[r].data_ptr() if sizes_want[r] else None
    # 1. a root buffer that is too small: every rank gets 0 and the sizes, nothing moves
    a = L.trico_hip_comm_gather(c, ptr, sizes_want[r], ROOT_RANK, out.data_ptr() if r == ROOT_RANK else None, 10 if r == ROOT_RANK else 0, sz)
    # 2. the real thing
    b = L.trico_hip_comm_gather(c, ptr, sizes_want[r], ROOT_RANK, out.data_ptr() if r == ROOT_RANK else None, total if r == ROOT_RANK else 0, sz)
    res[r] = (a, b, list(sz))
    L.trico_hip_comm_destroy(c)

th = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
for t in th: t.start()
for t in th: t.join(120)
assert all(not t.is_alive() for t in th), "a rank hangs"
for r in range(W):
    a, b, sz = res[r]
    assert a == 0 and b == 1 and sz == sizes_want, (r, res[r])
got = out.cpu().numpy()
want = np.concatenate([d.cpu().numpy() for d in data])
assert (got[:total] == want).all() and not got[total:].any()
print("FAKE GATHER OK")
"""


@pytest.mark.gpu
def test_comm_gather_offsets_with_four_ranks():
    env = dict(os.environ)
    env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FAKE GATHER OK" in out.stdout, out.stdout + out.stderr
