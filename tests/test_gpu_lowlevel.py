"""The reference's stand-alone coder and transpose entry points (floating_point_stream_compression.h,
transpose_aos_to_soa.h) as exported by libtrico.so, against the oracle / numpy on the same inputs."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(native_libs):
    lib = native_libs.lib()
    assert lib.trico_hip_available() == 1
    return ctypes.CDLL(native_libs.LIB_PATH)


def _data(kind, n, dt, rng):
    if kind == "smooth":
        return np.cumsum(rng.normal(0, 1e-3, n)).astype(dt)
    if kind == "noisy":
        return rng.normal(0, 1, n).astype(dt)
    return np.repeat(rng.normal(0, 1, n // 50 + 1), 50)[:n].astype(dt)


def _compress(L, vals, e1, e2):
    nb, out = ctypes.c_uint32(0), ctypes.c_void_p(None)
    if vals.dtype == np.float32:
        L.trico_compress(ctypes.byref(nb), ctypes.byref(out), vals.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(vals.size),
                         ctypes.c_uint32(e1), ctypes.c_uint32(e2))
    else:
        L.trico_compress_double_precision(ctypes.byref(nb), ctypes.byref(out), vals.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_uint32(vals.size), ctypes.c_uint64(e1), ctypes.c_uint64(e2))
    if not out.value:
        return None
    b = ctypes.string_at(out.value, nb.value)
    ctypes.CDLL(None).free(out)
    return b


def _decompress(L, payload, dt):
    n, out = ctypes.c_uint32(0), ctypes.c_void_p(None)
    buf = ctypes.create_string_buffer(payload, len(payload))
    f = L.trico_decompress if dt == np.float32 else L.trico_decompress_double_precision
    f(ctypes.byref(n), ctypes.byref(out), buf)
    if not out.value:
        return None
    a = np.frombuffer(ctypes.string_at(out.value, n.value * np.dtype(dt).itemsize), dt).copy()
    ctypes.CDLL(None).free(out)
    return a


@pytest.mark.parametrize("kind", ["smooth", "noisy", "steps"])
@pytest.mark.parametrize("n", [1, 9, 64, 1000, 70001])
def test_compress_decompress_default_exponents(L, kind, n):
    rng = np.random.default_rng(n)
    for dt, e in ((np.float32, (4, 10)), (np.float64, (20, 20))):
        vals = _data(kind, n, dt, rng)
        got = _compress(L, vals, *e)
        assert got == O.fpc_encode(vals, *e)
        back = _decompress(L, got, dt)
        assert back is not None and back.tobytes() == vals.tobytes()


@pytest.mark.parametrize("e", [(2, 2), (2, 10), (4, 6), (4, 8)])
def test_float_other_exponents(L, e):
    vals = _data("noisy", 5003, np.float32, np.random.default_rng(5))
    got = _compress(L, vals, *e)
    assert got == O.fpc_encode(vals, *e)
    assert _decompress(L, got, np.float32).tobytes() == vals.tobytes()


@pytest.mark.parametrize("e", [(2, 2), (10, 12), (16, 20), (20, 18)])
def test_double_other_exponents(L, e):
    vals = _data("smooth", 4099, np.float64, np.random.default_rng(6))
    got = _compress(L, vals, *e)
    assert got == O.fpc_encode(vals, *e)
    assert _decompress(L, got, np.float64).tobytes() == vals.tobytes()


@pytest.mark.parametrize("e", [(6, 10), (4, 12), (10, 12), (3, 11), (5, 10), (16, 22), (26, 24), (31, 40), (0, 10), (4, 0)])
def test_float_any_exponents_like_the_reference(L, e):
    """fpsc.c:88-93: any exponent is legal; odd ones are rounded down, everything above 30 becomes 30.  Tables beyond
    (4,10) live in global memory.  (31, 40) normalises to (30, 30): two 4 GiB tables."""
    vals = _data("noisy", 3001, np.float32, np.random.default_rng(sum(e)))
    got = _compress(L, vals, *e)
    assert got is not None
    assert got == O.fpc_encode(vals, *e)
    assert _decompress(L, got, np.float32).tobytes() == vals.tobytes()


@pytest.mark.parametrize("e", [(22, 20), (20, 24), (7, 9), (26, 26), (0, 20)])
def test_double_any_exponents_like_the_reference(L, e):
    vals = _data("steps", 2005, np.float64, np.random.default_rng(sum(e)))
    got = _compress(L, vals, *e)
    assert got is not None
    assert got == O.fpc_encode(vals, *e)
    assert _decompress(L, got, np.float64).tobytes() == vals.tobytes()


def _ptrs(arrs):
    boxes = [ctypes.c_void_p(a.ctypes.data) for a in arrs]
    return boxes, [ctypes.byref(b) for b in boxes]


@pytest.mark.parametrize("n", [1, 5, 257, 100003])
def test_real_transposes(L, n):
    rng = np.random.default_rng(n)
    for dt, sfx in ((np.float32, ""), (np.float64, "_double_precision")):
        for arity, name in ((3, "xyz"), (2, "uv")):
            aos = rng.normal(0, 1, n * arity).astype(dt)
            comps = [np.empty(n, dt) for _ in range(arity)]
            boxes, refs = _ptrs(comps)
            getattr(L, "trico_transpose_%s_aos_to_soa%s" % (name, sfx))(*refs, aos.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n))
            for c in range(arity):
                assert comps[c].tobytes() == aos.reshape(n, arity)[:, c].tobytes()
            back = np.empty_like(aos)
            bb = ctypes.c_void_p(back.ctypes.data)
            getattr(L, "trico_transpose_%s_soa_to_aos%s" % (name, sfx))(ctypes.byref(bb), *[c.ctypes.data_as(ctypes.c_void_p) for c in comps],
                                                                         ctypes.c_uint32(n))
            assert back.tobytes() == aos.tobytes()


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4099, 300001])
def test_integer_plane_transposes(L, n):
    rng = np.random.default_rng(n)
    for dt, bits in ((np.uint16, 16), (np.uint32, 32), (np.uint64, 64)):
        w = bits // 8
        vals = rng.integers(0, 1 << min(bits, 63), n, dtype=np.uint64).astype(dt)
        planes = [np.empty(n, np.uint8) for _ in range(w)]
        boxes, refs = _ptrs(planes)
        getattr(L, "trico_transpose_uint%d_aos_to_soa" % bits)(*refs, vals.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n))
        want = vals.view(np.uint8).reshape(n, w)
        for k in range(w):
            assert planes[k].tobytes() == want[:, k].tobytes(), (bits, k)
        back = np.empty_like(vals)
        bb = ctypes.c_void_p(back.ctypes.data)
        getattr(L, "trico_transpose_uint%d_soa_to_aos" % bits)(ctypes.byref(bb), *[p.ctypes.data_as(ctypes.c_void_p) for p in planes],
                                                               ctypes.c_uint32(n))
        assert back.tobytes() == vals.tobytes()
