import ctypes, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests")
import numpy as np
from trico_amd import api
L = api.lib()
n = 9 * (1 << 20) + 12345
rng = np.random.default_rng(1)
base = rng.integers(0, 256, 30011, dtype=np.uint8)
cases = {}
d = np.tile(base, n // base.size + 1)[:n].copy(); d[rng.integers(0, n, 40)] ^= 0x55; cases["period30k"] = d
parts, pos = [], 0
while pos < n:
    per = int(rng.integers(2000, 50000)); reps = int(rng.integers(3, 40))
    parts.append(np.tile(rng.integers(0, 256, per, dtype=np.uint8), reps)); pos += per * reps
cases["period_drift"] = np.concatenate(parts)[:n].copy()
for k, d in cases.items():
    a = api.Archive.open_for_writing(1 << 16)
    assert a.write("attributes_uint8", np.ascontiguousarray(d), n) == 1
    st = (ctypes.c_uint32 * 4)(); L.trico_hip_last_stats(st)
    print(k, "chunks accepted", st[0], "re-parsed or adopted", st[1], "archive bytes", a.get_size())
    a.close()
