"""Integer encoder (byte planes + LZ4) at small and middle sizes: the one-workgroup-per-plane compressor (k_lz4.hip) against the
chunk-speculative one (k_lz4_chunked.hip), which TRICO_LZ4_CHUNKED_MIN switches between (plane bytes).  Run once per setting."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

L = api.lib()
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 8)()
kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
for W, H in ((100, 100), (250, 200), (500, 350), (800, 500), (1000, 800), (1500, 1000), (2000, 1500)):
    _, t = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H)
    n = t.size
    d = torch.from_numpy(t.view(np.int32)).cuda()
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert L.trico_hip_int_encode(ctx, d.data_ptr(), n, 4, sizes) == 1, api.last_error()
        L.trico_hip_synchronize()
        best = min(best, time.perf_counter() - t0)
    pay = [torch.empty(sizes[c], dtype=torch.uint8, device="cuda") for c in range(4)]
    for c in range(4):
        assert L.trico_hip_fetch_payload(ctx, c, pay[c].data_ptr()) == 1
    out = torch.empty_like(d)
    pp = (ctypes.c_void_p * 8)(*[p.data_ptr() for p in pay] + [None] * 4)
    ss = (ctypes.c_uint32 * 8)(*[sizes[c] for c in range(4)] + [0] * 4)
    dbest = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert L.trico_hip_int_decode(ctx, pp, ss, 4, n, out.data_ptr()) == 1, api.last_error()
        L.trico_hip_synchronize()
        dbest = min(dbest, time.perf_counter() - t0)
    assert torch.equal(out, d)
    print("plane bytes %9d  encode %.3f ms  decode %.3f ms  sizes %s" % (n, best * 1e3, dbest * 1e3, list(sizes)[:4]), flush=True)
L.trico_hip_ctx_destroy(ctx)
