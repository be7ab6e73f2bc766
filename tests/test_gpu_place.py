"""GPU tier: a float stream coded AND framed in one queue of launches (trico_hip_fpc_encode_place, include/trico/trico_hip.h): what the
writers of csrc/host/archive.c do for a device-resident archive that has room for the stream's worst case.  Against the oracle's
payloads (fpsc.c:86-210) and the container's framing `u32 bytes, payload` per component (trico.c:215-262)."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,W,H,arity", [("walk", 300, 77, 3), ("grid", 1000, 1000, 3), ("walk", 1000, 1000, 2), ("grid", 640, 480, 1)])
def test_encode_place_frames_the_stream_body_like_the_reference(native_libs, kind, W, H, arity):
    native_libs.lib()
    import torch
    from trico_amd import api, meshgen
    L = api.lib()
    v, _ = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H, triangles=False)
    n = W * H
    data = np.ascontiguousarray(v.reshape(n, 3)[:, :arity])
    d = torch.from_numpy(data).cuda()
    bound = 5 + 4 * n + 3 * ((n + 7) // 8 + 1) + 8
    dst = torch.zeros(arity * (4 + bound) + 64, dtype=torch.uint8, device="cuda")
    ctx = L.trico_hip_ctx_create()
    sizes = (ctypes.c_uint32 * 3)()
    assert L.trico_hip_fpc_encode_place(ctx, d.data_ptr(), n, arity, 4, dst.data_ptr(), sizes) == 1, api.last_error()
    L.trico_hip_synchronize()
    got = dst.cpu().numpy()
    off = 0
    for c in range(arity):
        want = O.fpc_encode(np.ascontiguousarray(data[:, c]))
        assert sizes[c] == len(want), (c, sizes[c], len(want))
        assert int.from_bytes(got[off:off + 4].tobytes(), "little") == len(want)
        assert got[off + 4:off + 4 + len(want)].tobytes() == want, c
        off += 4 + len(want)
    assert not got[off:off + 64].any()                       # nothing behind the last payload
    # not this way: a host destination, an empty stream, doubles
    host = np.zeros(64, np.uint8)
    assert L.trico_hip_fpc_encode_place(ctx, d.data_ptr(), n, arity, 4, host.ctypes.data, sizes) == -1
    assert L.trico_hip_fpc_encode_place(ctx, d.data_ptr(), 0, arity, 4, dst.data_ptr(), sizes) == -1
    assert L.trico_hip_fpc_encode_place(ctx, d.data_ptr(), n // 2, arity, 8, dst.data_ptr(), sizes) == -1
    L.trico_hip_ctx_destroy(ctx)


@pytest.mark.parametrize("room", ["worst case", "tight"])
def test_device_archive_with_and_without_room_for_the_worst_case(native_libs, room):
    """The writer frames a float stream in place when the device archive has room for its worst case, and takes sizes first, payloads
    second (growing the buffer) when it has not: the same archive either way, the reference's."""
    native_libs.lib()
    from trico_amd import api
    from streams import mesh_streams
    for kind, W, H in (("walk", 300, 77), ("grid", 1000, 1000)):
        s = mesh_streams(kind, W, H)
        raw = sum(len(memoryview(data).cast("B")) for _, data, _ in s)
        a = api.Archive.open_for_writing(2 * raw + 4096 if room == "worst case" else 1 << 12, device=True)
        o = O.OracleArchive()
        for name, data, count in s:
            assert a.write(name, data, count) == 1, api.last_error()
            o.write(name, data, count)
        assert a.tobytes() == o.tobytes(), (kind, W, H, room)
        a.close(); o.close()


def test_in_place_framing_behind_the_two_sweep_coder():
    """The same framing when the slots come from the two-sweep coder (what a device whose LDS exchange is not trusted runs; here chosen
    with TRICO_FPC32_SWEEPS=2 in the test-hooks library): the gather is the same kernel, the slots have no records."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from oracle import oracle as O
from streams import mesh_streams
assert api.lib().trico_hip_fpc32_code_sweep() == 2
for kind, W, H in (("walk", 300, 77), ("grid", 1000, 1000)):
    s = mesh_streams(kind, W, H)
    raw = sum(len(memoryview(d).cast("B")) for _, d, _ in s)
    a = api.Archive.open_for_writing(2 * raw + 4096, device=True)
    o = O.OracleArchive()
    for name, data, count in s:
        assert a.write(name, data, count) == 1, api.last_error()
        o.write(name, data, count)
    assert a.tobytes() == o.tobytes(), (kind, W, H)
    a.close(); o.close()
print("PLACED")
"""
    env = dict(os.environ, TRICO_AMD_LIB=os.path.join(root, "tests", "_build", "libtrico_testhooks.so"), TRICO_FPC32_SWEEPS="2")
    out = subprocess.run([sys.executable, "-c", child % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "PLACED" in out.stdout, out.stdout + out.stderr
