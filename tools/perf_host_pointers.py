"""Config 2 through the plain C API with HOST pointers (PCIe staging included), for the note in DESIGN.md section 5."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trico_amd import api, meshgen

W, H = (10000, 5000) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
api.lib()
v, t = meshgen.grid(W, H)
nv, nt = W * H, 2 * W * H
raw = v.nbytes + t.nbytes
# the copies alone: the triangles (1.2 GB) up and down through trico_hip_copy (staging.hip for pageable memory of this size)
import ctypes
L = api.lib()
d = L.trico_hip_device_alloc(t.nbytes)
assert d
back = np.empty_like(t)
for it in range(3):
    c0 = time.perf_counter()
    assert L.trico_hip_copy(d, api.ptr(t), t.nbytes)
    c1 = time.perf_counter()
    assert L.trico_hip_copy(api.ptr(back), d, t.nbytes)
    c2 = time.perf_counter()
    print("copy %d: up %.1f ms (%.1f GB/s), down %.1f ms (%.1f GB/s)" % (it, (c1 - c0) * 1e3, t.nbytes / (c1 - c0) / 1e9,
                                                                       (c2 - c1) * 1e3, t.nbytes / (c2 - c1) / 1e9), flush=True)
assert back.tobytes() == t.tobytes()
L.trico_hip_device_free(d)
del back
for it in range(3):
    t0 = time.perf_counter()
    a = api.Archive.open_for_writing(raw // 4)
    assert a.write("vertices", v, nv) == 1 and a.write("triangles", t, nt) == 1
    t1 = time.perf_counter()
    blob = a.tobytes()
    a.close()
    v2, t2 = np.empty_like(v), np.empty_like(t)
    t2s = time.perf_counter()
    r = api.Archive.open_for_reading(blob)
    assert r.read("vertices", v2) == 1 and r.read("triangles", t2) == 1
    r.close()
    t3 = time.perf_counter()
    assert v2.tobytes() == v.tobytes() and t2.tobytes() == t.tobytes()
    print("iter %d: encode %.3f s (%.2f GB/s), decode %.3f s (%.2f GB/s), both %.3f GB/s" % (
        it, t1 - t0, raw / (t1 - t0) / 1e9, t3 - t2s, raw / (t3 - t2s) / 1e9, raw / (t1 - t0 + t3 - t2s) / 1e9), flush=True)
