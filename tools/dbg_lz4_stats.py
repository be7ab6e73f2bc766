"""Stitch statistics of the chunk-speculative LZ4 compressor on planes of a few hundred KB: chunks accepted / re-parsed serially."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trico_amd import api
from oracle import oracle as O
rng = np.random.default_rng(7)
cases = [("lowent3_300k", rng.integers(0, 3, 300000, dtype=np.uint8)), ("lowent3_1m", rng.integers(0, 3, 1000000, dtype=np.uint8)),
         ("lowent16_300k", rng.integers(0, 16, 300000, dtype=np.uint8)), ("walkish_300k", (np.cumsum(rng.integers(-2, 3, 300000)) & 255).astype(np.uint8)),
         ("random_300k", rng.integers(0, 256, 300000, dtype=np.uint8)), ("seq7_600k", np.tile(rng.integers(0, 256, (1, 7), dtype=np.uint8), (90000, 1)).reshape(-1)[:600000] ^ (rng.integers(0, 40, 600000) == 0).astype(np.uint8))]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from streams import mesh_streams
for W in (224, 500):
    nm, t, cnt = [x for x in mesh_streams("walk", W, W) if x[0] == "triangles"][0]
    w = api.Archive.open_for_writing(1 << 16)
    assert w.write(nm, t, cnt) == 1, api.last_error()
    got = w.tobytes(); w.close()
    st = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(st)
    o = O.OracleArchive(); o.write(nm, t, cnt); want = o.tobytes(); o.close()
    print("walk %d triangles: plane bytes %d exact=%s accepted=%d reparsed=%d" % (W, 3 * cnt, got == want, st[0], st[1]), flush=True)
for name, a in cases:
    a = np.ascontiguousarray(a)
    w = api.Archive.open_for_writing(1 << 16)
    assert w.write("attributes_uint8", a, a.size) == 1, api.last_error()
    got = w.tobytes(); w.close()
    st = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(st)
    o = O.OracleArchive(); o.write("attributes_uint8", a, a.size); want = o.tobytes(); o.close()
    print("%-14s n=%7d archive=%7d exact=%s accepted=%d reparsed=%d" % (name, a.size, len(got), got == want, st[0], st[1]), flush=True)
