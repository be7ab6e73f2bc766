"""Config 3 / 5 check at full size: multi(10000,5000) = 50M double vertices + double normals + float uv
(+ u64 triangles): device-resident encode, sha256 against the reference's golden, decode, timings."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from trico_amd import api, meshgen

W, H = (10000, 5000) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
with_tris = "--no-tris" not in sys.argv
L = api.lib()
t0 = time.perf_counter()
v, nrm, uv, t = meshgen.multi(W, H, triangles=with_tris)
print("generated in %.1f s" % (time.perf_counter() - t0), flush=True)
n = W * H
streams = [("vertices_double", v, n), ("vertex_normals_double", nrm, n), ("uv_per_vertex", uv, n)]
if with_tris:
    streams.append(("triangles_long", t, 2 * n))
dev = [(name, torch.from_numpy(a.view(np.uint8)).cuda(), cnt) for name, a, cnt in streams]
raw = sum(a.nbytes for _, a, _ in streams)
L.trico_hip_profile_enable(1)
a = api.Archive.open_for_writing(1 << 20, device=True)
for name, d, cnt in dev:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert a.write(name, d, cnt) == 1, api.last_error()
    torch.cuda.synchronize()
    print("write %-24s %8.1f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
size = a.get_size()
blob = a.tobytes()
sha = hashlib.sha256(blob).hexdigest()
print("archive bytes", size, "sha256", sha, flush=True)
hp = os.path.join(ROOT, "tests", "golden", "hashes.json")
g = json.load(open(hp)).get("multi_%dx%d" % (W, H))
if g and with_tris:
    print("golden match:", g["sha256"] == sha and g["size"] == size, flush=True)
r = api.Archive.open_for_reading(a.get_buffer_pointer(), size)
ok = True
for name, d, cnt in dev:
    out = torch.empty_like(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert r.read(name, out) == 1, api.last_error()
    torch.cuda.synchronize()
    same = bool(torch.equal(out, d))
    ok = ok and same
    print("read  %-24s %8.1f ms  exact=%s" % (name, (time.perf_counter() - t0) * 1e3, same), flush=True)
print("raw bytes", raw, "round trip exact:", ok)
