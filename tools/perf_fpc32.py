"""Micro-benchmark of the float-vertex encoder on device-resident input (not the driver bench)."""
import ctypes
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
W, H = (10000, 5000) if len(sys.argv) < 4 else (int(sys.argv[2]), int(sys.argv[3]))
L = api.lib()
v, _ = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H, triangles=False)
n = W * H
d = torch.from_numpy(v).cuda()
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 3)()
L.trico_hip_profile_enable(1)
place = os.environ.get("PERF_PLACE", "1") == "1"      # coded and framed in one queue of launches, as the archive writer does for a device archive
bound = 5 + 4 * n + 3 * ((n + 7) // 8 + 1) + 8
dst = torch.empty(3 * (4 + bound) + 1024, dtype=torch.uint8, device="cuda")
for it in range(6):
    if it == 1:
        L.trico_hip_profile_reset()
    if os.environ.get("PERF_IDLE_MS"):                  # an idle device in front of every encode (what a step of bench.py looks like)
        time.sleep(float(os.environ["PERF_IDLE_MS"]) / 1e3)
    t0 = time.perf_counter()
    if place:
        assert L.trico_hip_fpc_encode_place(ctx, d.data_ptr(), n, 3, 4, dst.data_ptr(), sizes) == 1, api.last_error()
    else:
        assert L.trico_hip_fpc_encode(ctx, d.data_ptr(), n, 3, 4, sizes) == 1, api.last_error()
        off = 0
        if os.environ.get("PERF_GATHER_ALL", "1") == "1":   # all payloads with ONE gather launch
            ptrs = (ctypes.c_void_p * 3)()
            for c in range(3):
                ptrs[c] = dst.data_ptr() + off
                off += sizes[c]
            assert L.trico_hip_fetch_payloads(ctx, 3, ptrs) == 1, api.last_error()
        else:
            for c in range(3):      # one launch per component
                assert L.trico_hip_fetch_payload(ctx, c, dst.data_ptr() + off) == 1, api.last_error()
                off += sizes[c]
    L.trico_hip_synchronize()
    t1 = time.perf_counter()
    print("iter", it, "wall ms %.3f" % ((t1 - t0) * 1e3), list(sizes), flush=True)
spans = ctypes.c_uint64(0)
ms = L.trico_hip_profile_ms(0, ctypes.byref(spans))
per = ms / spans.value
raw = n * 12
comp = sum(sizes)
print("kernel span avg %.3f ms; raw %.1f MB comp %.1f MB; input GB/s %.1f; algorithmic GB/s %.1f (%.1f%% of 8 TB/s)" % (
    per, raw / 1e6, comp / 1e6, raw / per / 1e6, (raw + comp) / per / 1e6, (raw + comp) / per / 1e6 / 8000 * 100))
L.trico_hip_ctx_destroy(ctx)
