"""Micro-benchmark of the integer (byte planes + LZ4) encoder/decoder on device-resident triangles."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
W, H = (10000, 5000) if len(sys.argv) < 4 else (int(sys.argv[2]), int(sys.argv[3]))
L = api.lib()
width = 8 if kind.endswith("64") else 4              # grid64 / walk64: the same indices as u64 (triangles_long)
_, t = (meshgen.grid if kind.startswith("grid") else meshgen.walk)(W, H)
nt = 2 * W * H
d = torch.from_numpy(t.astype(np.uint64).view(np.int64) if width == 8 else t.view(np.int32)).cuda()
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 8)()
st = (ctypes.c_uint32 * 4)()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert L.trico_hip_int_encode(ctx, d.data_ptr(), 3 * nt, width, sizes) == 1, api.last_error()
    L.trico_hip_synchronize()
    t1 = time.perf_counter()
    L.trico_hip_last_stats(st)
    print("encode iter", it, "wall ms %.2f" % ((t1 - t0) * 1e3), list(sizes)[:width], "chunks accepted", st[0], "reparsed", st[1], flush=True)
pay = [torch.empty(sizes[c], dtype=torch.uint8, device="cuda") for c in range(width)]
for c in range(width):
    assert L.trico_hip_fetch_payload(ctx, c, pay[c].data_ptr()) == 1
out = torch.empty_like(d)
pp = (ctypes.c_void_p * 8)(*[p.data_ptr() for p in pay] + [None] * (8 - width))
ss = (ctypes.c_uint32 * 8)(*[sizes[c] for c in range(width)] + [0] * (8 - width))
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert L.trico_hip_int_decode(ctx, pp, ss, width, 3 * nt, out.data_ptr()) == 1, api.last_error()
    L.trico_hip_synchronize()
    t1 = time.perf_counter()
    print("decode iter", it, "wall ms %.2f" % ((t1 - t0) * 1e3), "ok", bool(torch.equal(out, d)), flush=True)
