"""CPU tier, this container only: the restatement against the REAL reference compiled from
/root/reference into oracle/_ref (skipped where that library is absent)."""
import numpy as np
import pytest

from oracle import oracle as O
from streams import mesh_streams

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libtrico_ref.so not built")


def _ref_safe(a, payload_len):
    # the reference under-sizes its output buffer (fpsc.c:95, 585): only call it when it cannot overflow
    n = a.size
    mx = (4 * n + 3 * (n + 7) // 8 + (n & 7)) if a.dtype.itemsize == 4 else (8 * n + n // 2 + (n & 1))
    return payload_len <= mx


def test_fp_streams_match_reference():
    rng = np.random.default_rng(11)
    checked = 0
    for n in [5, 8, 9, 17, 63, 64, 65, 1000, 21845, 100003]:
        for dt in (np.float32, np.float64):
            for kind in range(4):
                if kind == 0:
                    a = np.cumsum(rng.standard_normal(n)).astype(dt) * 0.01
                elif kind == 1:
                    a = (np.arange(n) * 0.25).astype(dt)
                elif kind == 2:
                    a = (np.cumsum(rng.integers(-128, 128, n)) / 1024.0).astype(dt)
                else:
                    a = np.where(rng.random(n) < 0.5, 1.5, rng.standard_normal(n)).astype(dt)
                o = O.fpc_encode(a)
                if _ref_safe(a, len(o)):
                    assert O.ref_fpc_encode(a) == o, (n, dt, kind)
                    checked += 1
    assert checked > 40


@pytest.mark.parametrize("dt,e", [(np.float32, (2, 2)), (np.float32, (2, 10)), (np.float32, (4, 6)), (np.float32, (4, 8)),
                                  (np.float64, (2, 2)), (np.float64, (10, 12)), (np.float64, (16, 20)), (np.float64, (20, 18))])
def test_fp_streams_other_table_exponents(dt, e):
    """the exponent pairs the low-level API tests use on the GPU (tests/test_gpu_lowlevel.py), pinned here"""
    rng = np.random.default_rng(3)
    for n in (17, 1000, 5003):
        a = (np.cumsum(rng.integers(-128, 128, n)) / 1024.0).astype(dt)
        o = O.fpc_encode(a, *e)
        if _ref_safe(a, len(o)):
            assert O.ref_fpc_encode(a, *e) == o, (n, e)
        assert O.fpc_decode(o, dt).tobytes() == a.tobytes()


def test_lz4_matches_reference():
    rng = np.random.default_rng(12)
    for n in [0, 1, 4, 12, 13, 14, 20, 64, 100, 4096, 65535, 65546, 65547, 65548, 70000, 300000]:
        for kind in range(5):
            if kind == 0:
                a = rng.integers(0, 256, n, dtype=np.uint8)
            elif kind == 1:
                a = (rng.integers(0, 256, n, dtype=np.uint8) > 250).astype(np.uint8)
            elif kind == 2:
                a = (np.arange(n) % 251).astype(np.uint8)
            elif kind == 3:
                a = np.zeros(n, np.uint8)
            else:
                a = (np.arange(n) // 7 % 256).astype(np.uint8) ^ (rng.integers(0, 256, n, dtype=np.uint8) > 200)
            assert O.ref_lz4_compress(a) == O.lz4_compress(a), (n, kind)


@pytest.mark.parametrize("kind,W,H", [("grid", 300, 200), ("walk", 300, 200), ("multi", 200, 100)])
def test_archives_match_reference(kind, W, H, native_libs):
    streams = mesh_streams(kind, W, H)
    a, r = O.OracleArchive(), O.RefArchive()
    for name, data, count in streams:
        a.write(name, data, count)
        assert r.write(name, data, count) == 1
    assert a.tobytes() == r.tobytes()
    a.close()
    r.close()
