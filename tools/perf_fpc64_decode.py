"""Per-component timing of the double decoder: each component of the multi mesh's vertices as its own arity-1 stream."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

W, H = (10000, 1000) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
L = api.lib()
v = meshgen.multi(W, H)[0]
n = W * H
ctx = L.trico_hip_ctx_create()
for c in range(3):
    col = torch.from_numpy(np.ascontiguousarray(v.reshape(-1, 3)[:, c])).cuda()
    sizes = (ctypes.c_uint32 * 3)()
    assert L.trico_hip_fpc_encode(ctx, col.data_ptr(), n, 1, 8, sizes) == 1, api.last_error()
    pay = torch.empty(sizes[0] + 64, dtype=torch.uint8, device="cuda")
    assert L.trico_hip_fetch_payload(ctx, 0, pay.data_ptr()) == 1
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    pp = (ctypes.c_void_p * 3)(pay.data_ptr(), None, None)
    for it in range(2):
        L.trico_hip_synchronize()
        t0 = time.perf_counter()
        assert L.trico_hip_fpc_decode(ctx, pp, sizes, 1, 8, n, out.data_ptr()) == 1, api.last_error()
        L.trico_hip_synchronize()
        t1 = time.perf_counter()
    ok = bool((out.view(torch.int64) == col.view(torch.int64)).all())
    print("comp %d payload %d B (%.2f B/value)  decode %.1f ms  %.1f ns/value  ok=%s" % (
        c, sizes[0], sizes[0] / n, (t1 - t0) * 1e3, (t1 - t0) * 1e9 / n, ok), flush=True)
L.trico_hip_ctx_destroy(ctx)
