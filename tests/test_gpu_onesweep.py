"""The one-sweep float encoder (k_fpc32_sweep.hip: k_fpc32_sweep / k_fpc32_scanfix / k_fpc32_gather; the library's
choice on a device that passes the lane-order test) against the oracle: the archives have to be the reference's bytes on streams built
to stress the deferred values - a stream whose every DFCM class is new (a record per value at the start of every segment), exact hits
that are deferred (residual length 0: four unused bytes in a row), the coder's "never written" mark as a value and as a stride (the
stream is then coded again by the ballot coder: trico_hip_encode_stats), one- and two-component streams, streams shorter than a
segment - and, in the second half of the file, with its write-side guard provoked (tests/_build/libtrico_testhooks.so)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from oracle import oracle as O
assert api.lib().trico_hip_fpc32_code_sweep() == %(mode)d
rng = np.random.default_rng(7)
def bits(u):
    return np.ascontiguousarray(u.astype(np.uint32)).view(np.float32)
cases = []
# every value a new DFCM class for a long while: strides with distinct top bits
n = 300000
cases.append(("vertices", bits(np.cumsum(rng.integers(0, 1 << 31, n * 3, dtype=np.int64)) & 0xffffffff), n))
# long runs of exact FCM hits interrupted by jumps into new classes (deferred values of length 0 and 4)
v = np.repeat(rng.integers(0, 1 << 32, 4000, dtype=np.int64), 97)[: 120000 * 3]
cases.append(("vertices", bits(v), 120000))
# constant stream, and a ramp (one class, stride hits)
cases.append(("vertices", bits(np.full(90000 * 3, 0x3f800000, dtype=np.int64)), 90000))
cases.append(("vertex_normals", (np.arange(200001 * 3, dtype=np.float32) * 0.25), 200001))
# the sweep's "never written" mark (0x7fc0dead, fpc32_common.hpp: SENT) as a payload: as a value (FCM entries), as a stride (DFCM entries),
# alone and mixed with values of other classes, so that a class whose entry legitimately holds the sentinel is looked up again
SENT = 0x7fc0dead
m = 150000 * 3
cases.append(("vertices", bits(np.full(m, SENT, dtype=np.int64)), 150000))
cases.append(("vertex_normals", bits((np.arange(m, dtype=np.int64) * SENT) & 0xffffffff), 150000))
mix = rng.integers(0, 1 << 32, m, dtype=np.int64)
mix[::2] = SENT
cases.append(("vertices", bits(mix), 150000))
mix2 = np.cumsum(np.where(rng.integers(0, 3, m) == 0, SENT, rng.integers(0, 1 << 20, m))) & 0xffffffff
cases.append(("vertex_normals", bits(mix2), 150000))
# random streams of many shapes: few distinct values, random walks with rare jumps, pure noise
for seed in range(12):
    r2 = np.random.default_rng(100 + seed)
    nn = int(r2.integers(1, 400000))
    kind = seed %% 4
    if kind == 0:
        u = r2.integers(0, 1 << 32, nn * 3, dtype=np.int64)
    elif kind == 1:
        u = r2.choice(r2.integers(0, 1 << 32, 5, dtype=np.int64), nn * 3)
    elif kind == 2:
        u = np.cumsum(r2.integers(-1000, 1000, nn * 3)) & 0xffffffff
    else:
        u = (np.cumsum(r2.integers(0, 50, nn * 3)) + (r2.integers(0, 2000, nn * 3) == 0) * r2.integers(0, 1 << 31, nn * 3)) & 0xffffffff
    cases.append(("vertices" if seed %% 2 else "vertex_normals", bits(u), nn))
# two components and one component, lengths around the step and segment sizes
for nn in (1, 7, 63, 64, 65, 1023, 1024, 1025, 70001):
    cases.append(("uv_per_vertex", rng.standard_normal(nn * 2).astype(np.float32), nn))
    cases.append(("attributes_float", (rng.integers(0, 50, nn) * 0.5).astype(np.float32), nn))
a = api.Archive.open_for_writing(1 << 16)
o = O.OracleArchive()
for name, data, count in cases:
    assert a.write(name, data, count) == 1, (name, count, api.last_error())
    o.write(name, data, count)
x, y = a.tobytes(), o.tobytes()
assert len(x) == len(y), (len(x), len(y))
assert x == y, [i for i in range(len(x)) if x[i] != y[i]][:8]
# and back
r = api.Archive.open_for_reading(x)
for name, data, count in cases:
    if name == "attributes_float":
        got = r.read_alloc(name, count, np.float32)
        assert got is not None, (name, api.last_error())
    else:
        got = np.zeros_like(data)
        assert r.read(name, got) == 1, (name, api.last_error())
    assert got.tobytes() == data.tobytes(), name
print("ONESWEEP OK", len(x))
"""


@pytest.mark.parametrize("sweeps", ["1", "2"])
def test_one_sweep_encoder_writes_the_reference_bytes(sweeps):
    """sweeps = 2: the same streams through round 3's encoder (index sweep + code sweep with the exchange)."""
    env = dict(os.environ)
    if sweeps != "1":                    # (the switch exists in the test-hooks build only; "1" is the product library as it is)
        env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
        env["TRICO_FPC32_SWEEPS"] = sweeps
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "mode": 3 if sweeps == "1" else 2}], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "ONESWEEP OK" in out.stdout, out.stdout + out.stderr


GUARD_CHILD = r"""
import ctypes
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from oracle import oracle as O
from streams import mesh_streams
L = api.lib()
L.trico_hip_encode_stats.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
def stats():
    out = (ctypes.c_uint32 * 2)()
    L.trico_hip_encode_stats(out)
    return list(out)
assert L.trico_hip_fpc32_code_sweep() == 3
before = stats()
for kind, W, H in (("walk", 300, 77), ("grid", 256, 128), ("walk", 1000, 1000)):
    s = mesh_streams(kind, W, H)
    a = api.Archive.open_for_writing(1 << 16)
    o = O.OracleArchive()
    for name, data, count in s:
        assert a.write(name, data, count) == 1, api.last_error()
        o.write(name, data, count)
    assert a.tobytes() == o.tobytes(), (kind, W, H)
    a.close(); o.close()
after = stats()
L.trico_hip_encode_scan_recodes.restype = ctypes.c_uint32
print("GUARD", after[0] - before[0], after[1] - before[1], L.trico_hip_fpc32_code_sweep(), L.trico_hip_encode_scan_recodes())
"""


def _guard_child(env_add):
    env = dict(os.environ)
    env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
    env.update(env_add)
    out = subprocess.run([sys.executable, "-c", GUARD_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "GUARD" in out.stdout, out.stdout + out.stderr
    vals = [int(x) for x in out.stdout.split("GUARD")[1].split()[:4]]
    _guard_child.scan = vals[3]
    return vals[:3]


def test_write_side_guard_catches_an_exchange_out_of_lane_order():
    """TRICO_HIP_ENCODE_SABOTAGE=1 (test-hooks library only) hands every run start in the upper half of a wave something else than
    its predecessor left, as an LDS unit that does not apply an exchange in lane order would.  The sampled steps of the one-sweep
    coder notice (k_fpc32_sweep.hip: resolve_guarded), the stream is coded again by the ballot coder, the device is not asked again
    in this process - and every archive is still the reference's."""
    order, sentinel, mode = _guard_child({"TRICO_HIP_ENCODE_SABOTAGE": "1"})
    assert order >= 1 and mode == 0, (order, sentinel, mode)


def test_write_side_guard_is_quiet_without_sabotage():
    order, sentinel, mode = _guard_child({})
    assert order == 0 and sentinel == 0 and mode == 3 and _guard_child.scan == 0, (order, sentinel, mode, _guard_child.scan)


def test_a_scan_workgroup_that_never_publishes_is_waited_for_with_a_bound():
    """TRICO_HIP_ENCODE_SABOTAGE=2 (test-hooks library only): the first workgroup of k_fpc32_scanfix keeps the word the chunks behind it
    wait for.  Their waits run out (512 polls under the hook, 2^21 in the product), FLAG_SCAN reaches the host, the streams with more
    than one chunk are coded again by the two-sweep coder, the device stays trusted - and every archive is still the reference's."""
    order, sentinel, mode = _guard_child({"TRICO_HIP_ENCODE_SABOTAGE": "2"})
    assert order == 0 and sentinel == 0 and mode == 3 and _guard_child.scan >= 1, (order, sentinel, mode, _guard_child.scan)


def test_product_library_has_no_encode_sabotage_switch():
    env = dict(os.environ)
    env["TRICO_HIP_ENCODE_SABOTAGE"] = "1"
    out = subprocess.run([sys.executable, "-c", GUARD_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "GUARD 0 0 3" in out.stdout, out.stdout + out.stderr


def test_sentinel_payloads_are_coded_again_and_counted():
    """A stream that stores the coder's "never written" mark (as a value: FCM table; as a stride: DFCM table) raises FLAG_SENTINEL;
    the ballot coder takes over for that stream only and the bytes are the reference's (the CHILD's archive compare covers it; here:
    the counter moves and the device stays trusted)."""
    child = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(root)r)
from trico_amd import api
from oracle import oracle as O
L = api.lib()
L.trico_hip_encode_stats.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
out = (ctypes.c_uint32 * 2)()
v = np.full(150000 * 3, 0x7fc0dead, dtype=np.uint32).view(np.float32)
a = api.Archive.open_for_writing(1 << 16); o = O.OracleArchive()
assert a.write("vertices", v, 150000) == 1, api.last_error()
o.write("vertices", v, 150000)
assert a.tobytes() == o.tobytes()
L.trico_hip_encode_stats(out)
print("SENT", out[0], out[1], L.trico_hip_fpc32_code_sweep())
"""
    out = subprocess.run([sys.executable, "-c", child % {"root": ROOT}], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SENT 0 1 3" in out.stdout, out.stdout + out.stderr


VERIFY_TAIL = r"""
import ctypes
vs = (ctypes.c_uint64 * 3)()
api.lib().trico_hip_encode_verify_stats(vs)
print("VERIFY", vs[0], vs[1], vs[2])
"""


def test_full_encode_verification_agrees_on_every_value():
    """trico_hip_set_encode_verify / TRICO_HIP_ENCODE_VERIFY=1: every float stream of the stress archive above is coded a second time by
    the ballot coder and compared byte for byte on the device: the exchange coder and the ballot coder agree on 100 %% of the values
    (the write-side guard samples 0.04 %%), and the archive is still the reference's."""
    env = dict(os.environ)
    env["TRICO_HIP_ENCODE_VERIFY"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "mode": 3} + VERIFY_TAIL], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ONESWEEP OK" in out.stdout, out.stdout + out.stderr
    streams, values, differed = [int(x) for x in out.stdout.split("VERIFY")[1].split()[:3]]
    assert streams >= 38 and values > 5_000_000 and differed == 0, (streams, values, differed)


def test_full_encode_verification_catches_what_the_sampling_guard_is_told_to_ignore():
    """Test-hooks library: the exchange sabotaged (TRICO_HIP_ENCODE_SABOTAGE=1) AND the guard's flags ignored (TRICO_HIP_ENCODE_KEEP_FLAGGED):
    the one-sweep coder's wrong payload would reach the archive.  With the full verification on, the comparison with the ballot coder
    notices, the ballot coder's payload is written, and every archive is the reference's."""
    child = GUARD_CHILD + VERIFY_TAIL
    env = dict(os.environ)
    env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
    env.update({"TRICO_HIP_ENCODE_SABOTAGE": "1", "TRICO_HIP_ENCODE_KEEP_FLAGGED": "1", "TRICO_HIP_ENCODE_VERIFY": "1"})
    out = subprocess.run([sys.executable, "-c", child % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "VERIFY" in out.stdout, out.stdout + out.stderr
    streams, values, differed = [int(x) for x in out.stdout.split("VERIFY")[1].split()[:3]]
    assert streams >= 1 and differed >= 1, (streams, values, differed)
    assert "ENCODE VERIFICATION FAILED" in out.stderr

