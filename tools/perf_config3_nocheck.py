"""timing experiments on the double decoder that may produce wrong values: config 3's decode seconds, no comparison"""
import os, sys, time
os.environ.setdefault("TRICO_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "libtrico_testhooks.so"))   # TRICO_HIP_DECODE_CHECK exists in the test-hooks build only
os.environ["TRICO_HIP_DECODE_CHECK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trico_amd import api, meshgen
W, H = 10000, 5000
dev = torch.device("cuda:0")
v, nrm, uv, t = meshgen.multi(W, H)
n = W * H
streams = [("vertices_double", v, n), ("vertex_normals_double", nrm, n)]
devs = [(name, torch.from_numpy(a.view(np.uint8)).to(dev), cnt) for name, a, cnt in streams]
a = api.Archive.open_for_writing(1 << 30, device=True)
for name, d, cnt in devs:
    assert a.write(name, d, cnt) == 1
for rep in range(2):
    outs = [torch.empty_like(d) for _, d, _ in devs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
    for (name, d, cnt), o in zip(devs, outs):
        r.read(name, o)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("decode_s %.3f ok %s" % (t1 - t0, all(bool(torch.equal(o, d)) for (_, d, _), o in zip(devs, outs))))
    r.close()
