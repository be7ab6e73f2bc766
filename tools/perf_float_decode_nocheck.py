"""timing experiments on the float chain that may produce wrong values: decode seconds of the config-2 vertices, no self-check"""
import os, sys, time
os.environ.setdefault("TRICO_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "libtrico_testhooks.so"))   # TRICO_HIP_DECODE_CHECK exists in the test-hooks build only
os.environ["TRICO_HIP_DECODE_CHECK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trico_amd import api, meshgen
W, H = 10000, 5000
dev = torch.device("cuda:0")
v, t = meshgen.grid(W, H)
d = torch.from_numpy(v).to(dev)
a = api.Archive.open_for_writing(1 << 30, device=True)
assert a.write("vertices", d, W * H) == 1
for rep in range(3):
    o = torch.empty_like(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
    r.read("vertices", o)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("decode_s %.4f ns_per_value_z %.2f ok %s" % (t1 - t0, (t1 - t0) / (W * H) * 1e9, bool(torch.equal(o.view(torch.int32), d.view(torch.int32)))))
    r.close()
