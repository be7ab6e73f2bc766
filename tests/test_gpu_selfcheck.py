"""The chain decoders check themselves (trico_amd/csrc/hip/shim.hip): what they decoded is coded again on the device and compared
with the payload; a stream that fails is decoded again, the last time in reference order.  TRICO_HIP_DECODE_SABOTAGE=k damages
the output of the first k attempts on purpose."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from streams import mesh_streams, STREAM_TAG

L = api.lib()
stats = (ctypes.c_uint32 * 4)()
for kind, W, H in (("grid", 64, 32), ("multi", 64, 33), ("walk", 128, 64)):
    streams = mesh_streams(kind, W, H)
    a = api.Archive.open_for_writing(1 << 16)
    for name, data, count in streams:
        assert a.write(name, data, count) == 1, api.last_error()
    blob = a.tobytes()
    a.close()
    r = api.Archive.open_for_reading(blob)
    for name, data, count in streams:
        got = np.empty_like(data)
        assert r.read(name, got) == 1, (name, api.last_error())
        assert got.tobytes() == data.tobytes(), name
    r.close()
L.trico_hip_last_stats(stats)
print("REPEATS", stats[2])
"""


def run_child(sabotage):
    env = dict(os.environ)
    env["TRICO_HIP_DECODE_SABOTAGE"] = str(sabotage)
    # the switch exists only in the test build of the library (trico_amd/build.py); the product library ignores the variable
    env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("REPEATS")][0]
    return int(line.split()[1])


@pytest.mark.gpu
def test_no_repeats_when_nothing_goes_wrong():
    assert run_child(0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("sabotage", [1, 3])
def test_damaged_decodes_are_caught_and_repeated(sabotage):
    # grid: 1 float stream; multi: 2 double + 1 float (uv); walk: 1 float -> 5 checked streams, each repeated `sabotage` times
    # (the third repeat is the reference-order kernel, which is not sabotaged)
    assert run_child(sabotage) == 5 * sabotage
