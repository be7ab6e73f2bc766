"""Double encoder at small and middle sizes: the one-wave-per-component encoder (k_fpc64.hip) against the throughput encoder
(k_fpc64_sort.hip), which TRICO_FPC64_SORT_MIN switches between.  usage: perf_fpc64_small.py  (run once per setting of the variable)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api

L = api.lib()
ctx = L.trico_hip_ctx_create()
sizes = (ctypes.c_uint32 * 3)()
rng = np.random.default_rng(3)
for n in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 1048576):
    v = np.empty(3 * n)
    v[0::3] = np.arange(n) * 0.25
    v[1::3] = np.arange(n) // 100 * 0.25
    v[2::3] = rng.standard_normal(n)
    d = torch.from_numpy(v.view(np.int64)).cuda()
    best = 1e9
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert L.trico_hip_fpc_encode(ctx, d.data_ptr(), n, 3, 8, sizes) == 1, api.last_error()
        L.trico_hip_synchronize()
        best = min(best, time.perf_counter() - t0)
    print("n %8d  best %.3f ms  %.1f ns per value  sizes %s" % (n, best * 1e3, best * 1e9 / (3 * n), list(sizes)), flush=True)
L.trico_hip_ctx_destroy(ctx)
