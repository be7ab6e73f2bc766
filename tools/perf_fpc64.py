"""Micro-benchmark of the double encoder on device-resident input (not the driver bench): the two vec3 double streams of
multi(10000,5000) - vertices (x, y smooth, z noisy) and normals - coded one after the other, as the archive writer does."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

W, H = (10000, 5000) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
L = api.lib()
v, nrm, _, _ = meshgen.multi(W, H, triangles=False)
n = W * H
streams = [("vertices", torch.from_numpy(v.view(np.int64)).cuda()), ("normals", torch.from_numpy(nrm.view(np.int64)).cuda())]
del v, nrm
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 3)()
L.trico_hip_profile_enable(1)
comp = {}
for it in range(4):
    if it == 1:
        L.trico_hip_profile_reset()
    for name, d in streams:
        t0 = time.perf_counter()
        assert L.trico_hip_fpc_encode(ctx, d.data_ptr(), n, 3, 8, sizes) == 1, api.last_error()
        L.trico_hip_synchronize()
        comp[name] = sum(sizes)
        print("iter", it, name, "wall ms %.3f" % ((time.perf_counter() - t0) * 1e3), list(sizes), flush=True)
spans = ctypes.c_uint64(0)
ms = L.trico_hip_profile_ms(api.KERNEL_IDS["fpc64_encode"], ctypes.byref(spans))
per = ms / spans.value
raw = n * 24
c = sum(comp.values()) / 2
print("kernel span avg %.3f ms per vec3 double stream; raw %.1f MB comp %.1f MB; input GB/s %.1f; algorithmic GB/s %.1f (%.2f%% of 8 TB/s)" % (
    per, raw / 1e6, c / 1e6, raw / per / 1e6, (raw + c) / per / 1e6, (raw + c) / per / 1e6 / 8000 * 100))
L.trico_hip_ctx_destroy(ctx)
