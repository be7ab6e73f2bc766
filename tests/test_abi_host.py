"""CPU tier: the C-ABI library loads, exports every symbol include/*.h declares, and the host-side
container logic (header, peeks, sequencing, skip) behaves like the reference — no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from streams import ALL_ORDER, STREAM_TAG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("trico.h", "trico_hip.h"):
        text = open(os.path.join(ROOT, "include", "trico", h)).read()
        names += re.findall(r"TRICO_API[^;(]*?\b(trico_\w+)\s*\(", text)
    return names


def test_exports_every_declared_symbol(native_libs):
    L = ctypes.CDLL(native_libs.LIB_PATH)
    names = declared_symbols()
    assert len([n for n in names if not n.startswith("trico_hip_")]) == 54     # trico/trico.h:36-94
    for n in names:
        assert hasattr(L, n), n
    assert set(native_libs.API_SYMBOLS) <= set(names)
    assert set(native_libs.HIP_SYMBOLS) <= set(names)


def test_header_of_empty_archive(native_libs):
    # trico.tests/trico_compression.cpp:14-43 (test_header)
    a = native_libs.Archive.open_for_writing(1024)
    assert a.get_size() == 8
    b = a.tobytes()
    assert b == bytes.fromhex("5472636f00000000")
    a.close()
    r = native_libs.Archive.open_for_reading(b)
    assert r.get_version() == 0
    assert r.get_next_stream_type() == native_libs.trico_empty
    assert r.skip_next_stream() == 1
    r.close()


def test_open_rejects_bad_magic(native_libs):
    assert native_libs.Archive.open_for_reading(b"Trcx\0\0\0\0") is None
    assert native_libs.Archive.open_for_reading(b"Trc") is None


def test_sequencing_peeks_and_skip(native_libs, gold_dir, allstreams):
    """Walk the all-stream golden archive with skip only (NULL outputs: no decode, so no GPU needed)."""
    blob = open(os.path.join(gold_dir, "allstreams.trc"), "rb").read()
    r = native_libs.Archive.open_for_reading(blob)
    peeks = ["vertices", "triangles", "uvs", "normals", "colors", "attributes"]
    for name, div, peek in ALL_ORDER:
        assert r.get_next_stream_type() == STREAM_TAG[name], name
        want = allstreams[name].size // div
        if name == "uv_per_triangle":
            want *= 3                     # the stream stores 3*nr positions (trico.c:579)
        for p in peeks:
            assert r.get_number_of(p) == (want if p == peek else 0), (name, p)
        # a reader for a different stream type fails without consuming
        wrong = "triangles" if name != "triangles" else "vertices"
        assert r.read(wrong, None) == 0
        assert r.get_next_stream_type() == STREAM_TAG[name]
        assert r.skip_next_stream() == 1
    assert r.get_next_stream_type() == native_libs.trico_empty
    assert r.skip_next_stream() == 1
    r.close()


def test_truncated_archive_fails_cleanly(native_libs, gold_dir):
    blob = open(os.path.join(gold_dir, "grid_16x8.trc"), "rb").read()
    r = native_libs.Archive.open_for_reading(blob[:200])
    assert r.get_next_stream_type() == native_libs.trico_vertex_float_stream
    assert r.skip_next_stream() == 0          # payload runs past the end
    assert r.get_next_stream_type() == native_libs.trico_vertex_float_stream   # nothing consumed
    r.close()


def test_writers_fail_loudly_without_gpu(native_libs):
    L = native_libs.lib()
    if L.trico_hip_available():
        pytest.skip("GPU present")
    a = native_libs.Archive.open_for_writing(64)
    v = np.zeros(9, np.float32)
    assert a.write("vertices", v, 3) == 0
    assert "no CPU fallback" in native_libs.last_error()
    assert a.get_size() == 8
    a.close()
