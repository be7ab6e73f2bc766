"""Randomized soak of the HIP path against the oracle (run on the GPU box): many sizes / distributions / stream kinds.
usage: python tools/soak.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from trico_amd import api
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
api.lib()


def reals(kind, n, dt):
    if kind == 0:
        a = np.cumsum(rng.normal(0, 1e-3, n))
    elif kind == 1:
        a = rng.normal(0, 1, n)
    elif kind == 2:
        a = np.repeat(rng.normal(0, 1, n // 37 + 1), 37)[:n]
    elif kind == 3:
        a = np.sin(np.arange(n) * 0.01) * 100 + rng.normal(0, 1e-4, n)
    elif kind == 4:
        a = np.where(rng.random(n) < 0.01, rng.normal(0, 1e6, n), np.arange(n) * 0.25)
    elif kind == 5:
        a = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, 3.5]), n)
    elif kind == 6:
        bits = rng.integers(0, 1 << 32 if dt == np.float32 else 1 << 63, n, dtype=np.uint64)
        return bits.astype(np.uint32).view(np.float32) if dt == np.float32 else bits.view(np.float64)
    else:
        a = np.round(np.cumsum(rng.integers(-3, 4, n)) * 0.5)
    return a.astype(dt)


FP = [("vertices", 3, np.float32), ("vertices_double", 3, np.float64), ("uv_per_vertex", 2, np.float32),
      ("vertex_normals", 3, np.float32), ("attributes_float", 1, np.float32), ("attributes_double", 1, np.float64),
      ("triangle_normals_double", 3, np.float64)]
INT = [("triangles", 3, np.uint32), ("triangles_long", 3, np.uint64), ("vertex_colors", 1, np.uint32),
       ("attributes_uint8", 1, np.uint8), ("attributes_uint16", 1, np.uint16), ("attributes_uint64", 1, np.uint64)]

t0 = time.time()
for case in range(cases):
    streams = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 1000, 4097, 65536, 70001, int(rng.integers(1, int(os.environ.get("SOAK_MAXN", "300000"))))]))
        if os.environ.get("SOAK_BIG"):                  # middle-sized streams only: the thresholds between the small and the throughput kernels
            n = int(rng.integers(6000, int(os.environ.get("SOAK_MAXN", "300000"))))
        if rng.random() < 0.6:
            name, arity, dt = FP[int(rng.integers(len(FP)))]
            data = np.stack([reals(int(rng.integers(8)), n, dt) for _ in range(arity)], -1).reshape(-1)
        else:
            name, arity, dt = INT[int(rng.integers(len(INT)))]
            k = int(rng.integers(4))
            hi = [256, 70000, 1 << 31, 1 << 20][k]
            if k == 3:
                base = np.arange(n * arity) // 3
                data = (base + rng.integers(0, 50, n * arity)).astype(dt) if dt != np.uint8 else rng.integers(0, 256, n * arity).astype(dt)
            else:
                data = rng.integers(0, min(hi, int(np.iinfo(dt).max) + 1), n * arity).astype(dt)
        streams.append((name, np.ascontiguousarray(data), n))
    # where the archive lives: host memory; device memory without room (sizes first, payloads second, the buffer grows); device memory with
    # room for every stream's worst case (float streams are framed in place: trico_hip_fpc_encode_place)
    where = int(rng.integers(3)) if os.environ.get("SOAK_DEVICE") else 0
    room = 4096 + 2 * sum(len(d.tobytes()) + 64 for _, d, _ in streams)
    a = api.Archive.open_for_writing(1 << 12) if where == 0 else api.Archive.open_for_writing(1 << 12 if where == 1 else room, device=True)
    o = O.OracleArchive()
    for name, data, n in streams:
        assert a.write(name, data, n) == 1, (case, name, api.last_error())
        o.write(name, data, n)
    got, want = a.tobytes(), o.tobytes()
    a.close(); o.close()
    assert got == want, "case %d: archive differs (%s)" % (case, [(s[0], s[2]) for s in streams])
    r = api.Archive.open_for_reading(got)
    for name, data, n in streams:
        if name in ("attributes_float", "attributes_double"):
            back = r.read_alloc(name, n, data.dtype)
            assert back is not None
        else:
            back = np.empty_like(data)
            assert r.read(name, back) == 1, (case, name, api.last_error())
        assert back.tobytes() == data.tobytes(), "case %d: %s decodes differently" % (case, name)
    r.close()
    if case % 25 == 24:
        print("case", case + 1, "ok, %.0f s" % (time.time() - t0), flush=True)
import ctypes
vs = (ctypes.c_uint64 * 3)()
api.lib().trico_hip_encode_verify_stats(vs)
st = (ctypes.c_uint32 * 4)()
api.lib().trico_hip_last_stats(st)
print("soak passed:", cases, "cases, seed", seed, "| encode verification: %d streams, %d values, %d differed | decode repeats %d, other-writer streams %d"
      % (vs[0], vs[1], vs[2], st[2], st[3]))
