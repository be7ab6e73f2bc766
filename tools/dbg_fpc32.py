"""Debug helper: float encoder payloads vs the oracle, first differences per component."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen
from oracle import oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
W, H = (16, 8) if len(sys.argv) < 4 else (int(sys.argv[2]), int(sys.argv[3]))
L = api.lib()
v, _ = getattr(meshgen, kind)(W, H, triangles=False)
v = v.reshape(-1, 3)
n = v.shape[0]
d = torch.from_numpy(v.copy()).cuda()
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 3)()
assert L.trico_hip_fpc_encode(ctx, d.data_ptr(), n, 3, 4, sizes) == 1, api.last_error()
for c in range(3):
    want = oracle.fpc_encode(np.ascontiguousarray(v[:, c]), 4, 10)
    got = np.zeros(sizes[c], dtype=np.uint8)
    assert L.trico_hip_fetch_payload(ctx, c, got.ctypes.data) == 1
    want = np.frombuffer(bytes(want), dtype=np.uint8)
    m = min(len(want), len(got))
    diff = np.nonzero(want[:m] != got[:m])[0]
    print("comp", c, "len want/got", len(want), len(got), "ndiff", len(diff), "first", diff[:10])
    if len(diff):
        p = int(diff[0])
        print("  want", want[max(0, p - 8):p + 16].tolist())
        print("  got ", got[max(0, p - 8):p + 16].tolist())
L.trico_hip_ctx_destroy(ctx)
