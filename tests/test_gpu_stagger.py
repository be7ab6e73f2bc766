"""GPU tier: the float encoder's staggered segment lengths (fpc32_common.hpp: Stagger - segments of the workgroups that are dispatched
first are longer, the last ones shorter) at sizes where they are in force and nothing divides evenly: stream lengths that are not
multiples of a block of 512 values, of the class size or of anything else, for three, two and one components, against the oracle.
(The full-size tests cover 50 M values; everything below ~4.7 M values per component uses equal segments.)"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _stream(rng, n, arity, kind):
    m = n * arity
    if kind == "noisy":
        u = (np.cumsum(rng.integers(-4000, 4000, m)) + rng.integers(0, 64, m)) & 0xffffffff
    elif kind == "smooth":
        u = (np.arange(m, dtype=np.int64) * 37 + (rng.integers(0, 5000, m) == 0) * rng.integers(0, 1 << 30, m)) & 0xffffffff
    else:
        u = rng.integers(0, 1 << 32, m, dtype=np.int64)
    return np.ascontiguousarray(u.astype(np.uint32)).view(np.float32)


@pytest.mark.parametrize("name,n,arity,kind", [
    ("vertices", 6_000_001, 3, "noisy"),
    ("vertex_normals", 9_999_937, 3, "smooth"),
    ("uv_per_vertex", 7_340_033, 2, "noisy"),
    ("attributes_float", 12_582_917, 1, "random"),
])
def test_staggered_segments_write_the_reference_bytes(native_libs, name, n, arity, kind):
    api = native_libs
    assert api.lib().trico_hip_available() == 1, api.last_error()
    rng = np.random.default_rng(n)
    data = _stream(rng, n, arity, kind)
    a = api.Archive.open_for_writing(1 << 20)
    assert a.write(name, data, n) == 1, api.last_error()
    got = a.tobytes()
    a.close()
    o = O.OracleArchive()
    o.write(name, data, n)
    want = o.tobytes()
    o.close()
    assert len(got) == len(want)
    assert got == want
    r = api.Archive.open_for_reading(got)
    if name == "attributes_float":
        back = r.read_alloc(name, n, np.float32)
        assert back is not None, api.last_error()
    else:
        back = np.zeros_like(data)
        assert r.read(name, back) == 1, api.last_error()
    r.close()
    assert back.tobytes() == data.tobytes()
