"""Large copies between pageable host memory and the device (trico_amd/csrc/hip/staging.hip): the ring of pinned chunks must move
exactly the bytes the runtime's own copy moves, for sizes around its chunk (16 MiB) and threshold (8 MiB), with and without helper
threads, and a process that used it must still exit (its threads sleep on a condition variable for the life of the process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIB = 1 << 20

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from trico_amd import api
L = api.lib()
rng = np.random.default_rng(5)
for nbytes in (8 * %d - 1, 8 * %d, 16 * %d + 4097, 70 * %d + 13):
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    d = L.trico_hip_device_alloc(nbytes + 64)
    assert d
    assert L.trico_hip_copy(d + 32, api.ptr(src), nbytes) == 1          # (an odd device offset too)
    back = np.zeros(nbytes + 2, dtype=np.uint8)
    assert L.trico_hip_copy(api.ptr(back) + 1, d + 32, nbytes) == 1
    assert back[0] == 0 and back[-1] == 0 and np.array_equal(back[1:-1], src), nbytes
    L.trico_hip_device_free(d)
print("ok")
""" % (ROOT, MIB, MIB, MIB, MIB)


@pytest.mark.parametrize("threads", ["0", "2", None], ids=["runtime", "two_threads", "default"])
def test_copies_are_exact_and_the_process_exits(threads):
    env = dict(os.environ)
    env.pop("TRICO_HIP_STAGE_THREADS", None)
    if threads is not None:
        env["TRICO_HIP_STAGE_THREADS"] = threads
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_archive_round_trip_through_host_arrays(native_libs):
    """A mesh big enough for the ring (vertices 24 MB, triangles 48 MB) written from and read into host arrays: the archive is the one
    the device-pointer path writes (tests/test_gpu_api.py pins that one against the oracle)."""
    from trico_amd import api, meshgen
    v, t = meshgen.walk(2000, 1000)
    a = api.Archive.open_for_writing(1 << 20)
    assert a.write("vertices", v, len(v) // 3) == 1
    assert a.write("triangles", t, len(t) // 3) == 1
    blob = a.tobytes()
    a.close()
    v2, t2 = np.empty_like(v), np.empty_like(t)
    r = api.Archive.open_for_reading(blob)
    assert r.read("vertices", v2) == 1 and r.read("triangles", t2) == 1
    r.close()
    assert v2.tobytes() == v.tobytes() and t2.tobytes() == t.tobytes()
