"""GPU tier: LZ4_compress_default / LZ4_decompress_safe / LZ4_compressBound of include/lz4/lz4.h (the entry points of the reference's
lz4.h:127-171 that trico.c:339-514, 1100-1129 and trico.tests/int_compression.cpp:75-187 call) on the HIP codec, against the oracle's
restatement of LZ4 1.9.2 (pinned to the compiled reference by tests/test_oracle_vs_reference.py)."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _lib(native_libs):
    native_libs.lib()                      # (imports torch first where it is installed: one HIP runtime per process)
    L = ctypes.CDLL(native_libs.LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.LZ4_compress_default.argtypes = [vp, vp, ci, ci]
    L.LZ4_compress_default.restype = ci
    L.LZ4_decompress_safe.argtypes = [vp, vp, ci, ci]
    L.LZ4_decompress_safe.restype = ci
    L.LZ4_compressBound.argtypes = [ci]
    L.LZ4_compressBound.restype = ci
    return L


def _cases():
    rng = np.random.default_rng(11)
    yield "empty", np.zeros(0, np.uint8)
    yield "one", np.array([7], np.uint8)
    yield "twelve", np.arange(12, dtype=np.uint8)
    yield "thirteen_zeros", np.zeros(13, np.uint8)
    yield "random_4k", rng.integers(0, 256, 4096, dtype=np.uint8)
    yield "ramp_70000", (np.arange(70000) // 7 % 256).astype(np.uint8)            # beyond the 65547-byte table switch (lz4.c:570)
    yield "lowentropy_300k", rng.integers(0, 3, 300000, dtype=np.uint8)            # the chunked compressor, short sequences
    yield "runs_2m", np.repeat(rng.integers(0, 256, 8000, dtype=np.uint8), 256)    # long matches, the data-parallel decoder


@pytest.mark.parametrize("name,data", list(_cases()), ids=[n for n, _ in _cases()])
def test_block_api_matches_lz4_1_9_2(native_libs, name, data):
    L = _lib(native_libs)
    want = O.lz4_compress(data)
    bound = L.LZ4_compressBound(data.size)
    assert bound == data.size + data.size // 255 + 16
    dst = np.full(bound + 8, 0xAB, np.uint8)
    n = L.LZ4_compress_default(data.ctypes.data if data.size else None, dst.ctypes.data, data.size, bound)
    assert n == len(want), native_libs.last_error()
    assert dst[:n].tobytes() == want
    assert (dst[bound:] == 0xAB).all()
    # a destination that is one byte short: the reference's output-limited variant returns 0 exactly then
    if len(want) > 1:
        assert L.LZ4_compress_default(data.ctypes.data, dst.ctypes.data, data.size, len(want) - 1) == 0
        assert L.LZ4_compress_default(data.ctypes.data, dst.ctypes.data, data.size, len(want)) == len(want)
    # back: exact capacity (what the archive format does, trico.c:1100-1129), then a capacity that is only an upper bound
    blk = np.frombuffer(want, np.uint8).copy()
    out = np.full(data.size + 64, 0xCD, np.uint8)
    assert L.LZ4_decompress_safe(blk.ctypes.data, out.ctypes.data, blk.size, data.size) == data.size, native_libs.last_error()
    assert out[:data.size].tobytes() == data.tobytes() and (out[data.size:] == 0xCD).all()
    out[:] = 0xCD
    assert L.LZ4_decompress_safe(blk.ctypes.data, out.ctypes.data, blk.size, data.size + 40) == data.size, native_libs.last_error()
    assert out[:data.size].tobytes() == data.tobytes() and (out[data.size:] == 0xCD).all()
    # too little room, and a damaged block: negative, nothing written beyond the capacity
    if data.size > 1:
        out[:] = 0xCD
        assert L.LZ4_decompress_safe(blk.ctypes.data, out.ctypes.data, blk.size, data.size - 1) < 0
        assert (out[data.size - 1:] == 0xCD).all()
    if blk.size > 20:
        bad = blk.copy()
        bad = bad[:blk.size - 3]                       # a block cut short ends inside its last literals
        assert L.LZ4_decompress_safe(bad.ctypes.data, out.ctypes.data, bad.size, data.size) < 0


def test_block_api_against_the_compiled_reference(native_libs):
    if not O.have_ref():
        pytest.skip("oracle/_ref/libtrico_ref.so not built")
    R = O.ref()
    L = _lib(native_libs)
    rng = np.random.default_rng(5)
    data = np.concatenate([rng.integers(0, 256, 3000, dtype=np.uint8), np.zeros(90000, np.uint8), rng.integers(0, 4, 50000, dtype=np.uint8)])
    bound = R.LZ4_compressBound(data.size)
    assert bound == L.LZ4_compressBound(data.size)
    a, b = np.zeros(bound, np.uint8), np.zeros(bound, np.uint8)
    na = R.LZ4_compress_default(data.ctypes.data, a.ctypes.data, data.size, bound)
    nb = L.LZ4_compress_default(data.ctypes.data, b.ctypes.data, data.size, bound)
    assert na == nb and a[:na].tobytes() == b[:nb].tobytes()
    for cap in (na - 1, na, na + 1):                   # the output-limited path of the reference (lz4.c:975-980, 1057-1062, 1153-1160)
        assert (R.LZ4_compress_default(data.ctypes.data, a.ctypes.data, data.size, cap) > 0) == (L.LZ4_compress_default(data.ctypes.data, b.ctypes.data, data.size, cap) > 0)


def test_block_api_takes_device_pointers(native_libs):
    """src / dst in HBM (the boundary's MI355X extension: any data pointer may be a HIP device pointer)"""
    import torch
    L = _lib(native_libs)
    rng = np.random.default_rng(3)
    data = np.concatenate([rng.integers(0, 4, 400000, dtype=np.uint8), np.zeros(700000, np.uint8), rng.integers(0, 256, 5000, dtype=np.uint8)])
    want = O.lz4_compress(data)
    d_src = torch.from_numpy(data).cuda()
    bound = L.LZ4_compressBound(data.size)
    d_dst = torch.zeros(bound, dtype=torch.uint8, device="cuda")
    n = L.LZ4_compress_default(d_src.data_ptr(), d_dst.data_ptr(), data.size, bound)
    assert n == len(want), native_libs.last_error()
    assert d_dst[:n].cpu().numpy().tobytes() == want
    d_out = torch.full((data.size + 32,), 0xEE, dtype=torch.uint8, device="cuda")
    assert L.LZ4_decompress_safe(d_dst.data_ptr(), d_out.data_ptr(), n, data.size) == data.size, native_libs.last_error()
    back = d_out.cpu().numpy()
    assert back[:data.size].tobytes() == data.tobytes() and (back[data.size:] == 0xEE).all()
    assert L.LZ4_decompress_safe(d_dst.data_ptr(), d_out.data_ptr(), n, data.size + 32) == data.size, native_libs.last_error()
