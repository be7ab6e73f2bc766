"""GPU tier, BASELINE.json's full sizes: device-resident inputs, archive sha256 against the goldens the compiled
reference produced (tests/golden/hashes.json), decode back and compare bit-exactly."""
import hashlib

import numpy as np
import pytest

from streams import mesh_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(native_libs):
    assert native_libs.lib().trico_hip_available() == 1, native_libs.last_error()
    return native_libs


def _device_streams(streams):
    import torch
    return [(name, torch.from_numpy(a.view(np.uint8)).cuda(), cnt) for name, a, cnt in streams]


def _repeats(api):
    """decodes this thread had to repeat because the self-check of the chain decoders failed (trico_hip_last_stats word 2)"""
    import ctypes
    st = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(st)
    return int(st[2])


def test_config2_full_size(api, hashes):
    """configs[1]: 50M float vertices + 100M uint32 triangles on one MI355X, bit-exact .trc vs the reference."""
    import torch
    dev = _device_streams(mesh_streams("grid", 10000, 5000))
    a = api.Archive.open_for_writing(1 << 20, device=True)
    for name, d, cnt in dev:
        assert a.write(name, d, cnt) == 1, api.last_error()
    blob = a.tobytes()
    g = hashes["grid_10000x5000"]
    assert len(blob) == g["size"]
    assert hashlib.sha256(blob).hexdigest() == g["sha256"]
    before = _repeats(api)
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
    for name, d, cnt in dev:
        out = torch.empty_like(d)
        assert r.read(name, out) == 1, api.last_error()
        assert torch.equal(out, d), name
    r.close()
    a.close()
    assert _repeats(api) == before            # the chains were right by themselves: the self-check had nothing to repeat


def test_config2_vertices_full_size_with_full_encode_verification(api, hashes):
    """The same 50 M vertices with trico_hip_set_encode_verify(1): all 150 M values coded a second time by the ballot coder and compared
    byte for byte on the device - the exchange coder and the ballot coder agree on every one of them (the write-side
    guard alone samples 0.04 %)."""
    import ctypes
    L = api.lib()
    L.trico_hip_encode_verify_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    before = (ctypes.c_uint64 * 3)()
    after = (ctypes.c_uint64 * 3)()
    L.trico_hip_encode_verify_stats(before)
    dev = _device_streams(mesh_streams("grid", 10000, 5000))
    L.trico_hip_set_encode_verify(1)
    try:
        a = api.Archive.open_for_writing(1 << 20, device=True)
        for name, d, cnt in dev:
            assert a.write(name, d, cnt) == 1, api.last_error()
        blob = a.tobytes()
        a.close()
    finally:
        L.trico_hip_set_encode_verify(-1)
    L.trico_hip_encode_verify_stats(after)
    g = hashes["grid_10000x5000"]
    assert len(blob) == g["size"] and hashlib.sha256(blob).hexdigest() == g["sha256"]
    assert after[0] - before[0] == 1 and after[1] - before[1] == 150_000_000 and after[2] == before[2]


class _Full:
    """One full-size archive kept on the device for the tests below: (device inputs, archive handle)."""
    cache = {}

    @classmethod
    def get(cls, api, kind):
        if kind not in cls.cache:
            dev = _device_streams(mesh_streams(kind, 10000, 5000))
            a = api.Archive.open_for_writing(1 << 20, device=True)
            for name, d, cnt in dev:
                assert a.write(name, d, cnt) == 1, api.last_error()
            cls.cache[kind] = (dev, a)
        return cls.cache[kind]

    @classmethod
    def drop(cls):
        for _, a in cls.cache.values():
            a.close()
        cls.cache.clear()


@pytest.fixture(scope="module")
def full(api):
    yield lambda kind: _Full.get(api, kind)
    _Full.drop()


def _decode_all(api, dev, a, errs, tag):
    import torch
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
    for name, d, cnt in dev:
        out = torch.empty_like(d)
        if r.read(name, out) != 1:
            errs.append((tag, name, api.last_error()))
            break
        if not torch.equal(out, d):
            errs.append((tag, name, "decoded bytes differ from the input"))
        del out
    if not errs and r.get_next_stream_type() != api.trico_empty:
        errs.append((tag, "end", "archive not exhausted"))
    r.close()


def test_config3_full_size(api, hashes, full):
    """configs[2]: 50M double vertices + double normals + float uv (+ uint64 triangles): archive sha256 vs the
    reference, then every stream (both double streams included) decoded back and compared bit for bit."""
    dev, a = full("multi")
    blob = a.tobytes()
    g = hashes["multi_10000x5000"]
    assert len(blob) == g["size"]
    assert hashlib.sha256(blob).hexdigest() == g["sha256"]
    del blob
    errs = []
    before = _repeats(api)
    _decode_all(api, dev, a, errs, "multi")
    assert not errs, errs
    assert _repeats(api) == before            # the double chains were right by themselves


@pytest.mark.parametrize("k", range(1, 8))
def test_config4_mesh_of_rank(api, hashes, k):
    """configs[3]: the mesh of rank k of the 8-GPU job (grid(10000,5000), seed 0x12345678 + k) encodes to the archive the
    reference writes for it (trico.tests/trico_compression.cpp:110-177 round trip, here pinned by sha256)."""
    seed = 0x12345678 + k
    dev = _device_streams(mesh_streams("grid", 10000, 5000, seed))
    a = api.Archive.open_for_writing(450 << 20, device=True)
    for name, d, cnt in dev:
        assert a.write(name, d, cnt) == 1, api.last_error()
    blob = a.tobytes()
    a.close()
    g = hashes["grid_10000x5000_seed%08x" % seed]
    assert len(blob) == g["size"]
    assert hashlib.sha256(blob).hexdigest() == g["sha256"]


def test_config5_concurrent_decode_of_mixed_archives(api, hashes, full):
    """configs[4]: eight archives with float, double and uint64 streams (grid, walk, multi at 10000x5000) decoded at the
    same time from eight host threads on one GPU; every decoded stream equals its input bit for bit, and the walk
    archive is the one the reference writes."""
    import threading
    kinds = ["grid", "walk", "multi", "grid", "walk", "grid", "walk", "grid"]
    blob = full("walk")[1].tobytes()
    g = hashes["walk_10000x5000"]
    assert len(blob) == g["size"] and hashlib.sha256(blob).hexdigest() == g["sha256"]
    del blob
    errs = []
    th = [threading.Thread(target=_decode_all, args=(api,) + full(k) + (errs, "%s#%d" % (k, i))) for i, k in enumerate(kinds)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
