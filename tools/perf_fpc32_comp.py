"""Float encoder per component: each component of a mesh encoded as its own arity-1 stream (kernel spans)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
W, H = (10000, 5000)
L = api.lib()
v, _ = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H, triangles=False)
v = v.reshape(-1, 3)
n = W * H
ctx = L.trico_hip_ctx_create()
L.trico_hip_profile_enable(1)
for c in range(3):
    col = torch.from_numpy(np.ascontiguousarray(v[:, c])).cuda()
    sizes = (ctypes.c_uint32 * 3)()
    dst = None
    for it in range(4):
        if it == 1:
            L.trico_hip_profile_reset()
        assert L.trico_hip_fpc_encode(ctx, col.data_ptr(), n, 1, 4, sizes) == 1, api.last_error()
        if dst is None:
            dst = torch.empty(sizes[0] + 1024, dtype=torch.uint8, device="cuda")
        assert L.trico_hip_fetch_payload(ctx, 0, dst.data_ptr()) == 1
        L.trico_hip_synchronize()
    spans = ctypes.c_uint64(0)
    ms = L.trico_hip_profile_ms(0, ctypes.byref(spans))
    print("comp %d: %.1f B/value, encoder span %.3f ms" % (c, sizes[0] / n, ms / spans.value), flush=True)
L.trico_hip_ctx_destroy(ctx)
