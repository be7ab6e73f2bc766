"""Debugging aid for the one-sweep float encoder: codes streams of several kinds with the hooks library, keeps what the sweep wrote even
when its guard raised a flag (TRICO_HIP_ENCODE_KEEP_FLAGGED), and says where the payload first differs from the oracle's."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("DBG_PRODUCT") != "1":
    os.environ.setdefault("TRICO_AMD_LIB", os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so"))
os.environ["TRICO_HIP_ENCODE_KEEP_FLAGGED"] = "1"
import numpy as np
import torch
from trico_amd import api, meshgen
from oracle import oracle as O

L = api.lib()
L.trico_hip_encode_stats.argtypes = [ctypes.POINTER(ctypes.c_uint32)]


def stats():
    out = (ctypes.c_uint32 * 2)()
    L.trico_hip_encode_stats(out)
    return list(out)


def bits(u):
    return np.ascontiguousarray(u.astype(np.uint32)).view(np.float32)


def locate(payload, off):
    """which group / value of a float payload byte `off` belongs to: walks the group headers (fpsc.c:12-74)"""
    pos = 5
    g = 0
    while pos < len(payload):
        h = (payload[pos] << 16) | (payload[pos + 1] << 8) | payload[pos + 2]
        codes = [(h >> (3 * j)) & 7 for j in range(8)]
        lens = [c if c <= 4 else c - 4 for c in codes]
        end = pos + 3 + sum(lens)
        if off < end:
            if off < pos + 3:
                return "header byte %d of group %d (values %d..%d, step %d, lanes %d..), codes %s" % (off - pos, g, 8 * g, 8 * g + 7, g // 8, (8 * g) % 64, codes)
            q = pos + 3
            for j in range(8):
                if off < q + lens[j]:
                    return "byte %d of the residual of value %d (step %d lane %d, code %d), group codes %s" % (off - q, 8 * g + j, (8 * g + j) // 64, (8 * g + j) % 64, codes[j], codes)
                q += lens[j]
        pos = end
        g += 1
    return "beyond the payload"


def check(name, v, arity):
    n = v.size // arity
    d = torch.from_numpy(v.view(np.uint32).astype(np.int64).astype(np.uint32).view(np.int32).copy()).cuda()
    ctx = L.trico_hip_ctx_create()
    sizes = (ctypes.c_uint32 * 3)()
    b = stats()
    assert L.trico_hip_fpc_encode(ctx, d.data_ptr(), n, arity, 4, sizes) == 1, api.last_error()
    a = stats()
    bad = False
    for c in range(arity):
        want = O.fpc_encode(np.ascontiguousarray(v.reshape(n, arity)[:, c]))
        got = torch.empty(max(int(sizes[c]), 1), dtype=torch.uint8, device="cuda")
        assert L.trico_hip_fetch_payload(ctx, c, got.data_ptr()) == 1, api.last_error()
        L.trico_hip_synchronize()
        g = got.cpu().numpy()[: sizes[c]].tobytes()
        if g != want:
            bad = True
            m = min(len(g), len(want))
            first = next((i for i in range(m) if g[i] != want[i]), m)
            print("  %s c%d: size %d (want %d), first difference at byte %d of the payload: got %s want %s" % (
                name, c, len(g), len(want), first, g[first:first + 12].hex(), want[first:first + 12].hex()))
            print("     want: " + locate(want, first))
            print("     got : " + locate(g, first))
            ndiff = sum(1 for i in range(m) if g[i] != want[i])
            print("     %d of %d bytes differ" % (ndiff, m))
    print("%-28s n %8d arity %d: %s; guard order +%d sentinel +%d" % (name, n, arity, "DIFFERENT" if bad else "ok", a[0] - b[0], a[1] - b[1]), flush=True)
    L.trico_hip_ctx_destroy(ctx)


rng = np.random.default_rng(5)
for n in (1024, 23100):
    check("const", bits(np.full(n * 3, 0x3f800000, dtype=np.int64)), 3)
    check("ramp", np.arange(n * 3, dtype=np.float32) * 0.25, 3)
    check("noise", bits(rng.integers(0, 1 << 32, n * 3, dtype=np.int64)), 3)
    check("walkbits", bits(np.cumsum(rng.integers(-1000, 1000, n * 3)) & 0xffffffff), 3)
    check("noise arity 1", bits(rng.integers(0, 1 << 32, n, dtype=np.int64)), 1)
for W, H in ((300, 77),):
    for kind in ("grid", "walk"):
        v, _ = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H, triangles=False)
        check("%s %dx%d" % (kind, W, H), v, 3)
