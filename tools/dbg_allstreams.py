import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trico_amd import api
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from streams import ALL_ORDER
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
d = np.load(os.path.join(G, "allstreams.npz"))
blob = open(os.path.join(G, "allstreams.trc"), "rb").read()
r = api.Archive.open_for_reading(blob)
for name, div, _ in ALL_ORDER:
    data = d[name]
    print("reading", name, data.size, flush=True)
    if name in ("attributes_float", "attributes_double"):
        got = r.read_alloc(name, data.size // div, data.dtype)
    else:
        got = np.empty_like(data)
        assert r.read(name, got) == 1, api.last_error()
    print("   ok" if got.tobytes() == data.tobytes() else "   MISMATCH", flush=True)
