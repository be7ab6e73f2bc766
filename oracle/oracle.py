"""ctypes access to the oracle libraries.  TEST INFRASTRUCTURE ONLY (see trico_oracle.c header):
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product package.

  liboracle.so          CPU restatement (built from oracle/trico_oracle.c by oracle/Makefile)
  _ref/libtrico_ref.so  the real reference compiled from /root/reference (present when built there;
                        travels to the GPU box prebuilt)
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libtrico_ref.so")

_orc = None
_ref = None


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def orc():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "trico_oracle.c")):
            build()
        L = ctypes.CDLL(ORACLE_SO)
        vp, u32, u64, ci, cu = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint
        L.orc_fpc_bound.argtypes = [u32, ci]; L.orc_fpc_bound.restype = u64
        L.orc_fpc_encode.argtypes = [vp, u32, ci, cu, cu, vp]; L.orc_fpc_encode.restype = u32
        L.orc_fpc_decode.argtypes = [vp, u64, ci, vp, u32, ctypes.POINTER(u32)]; L.orc_fpc_decode.restype = ci
        L.orc_deinterleave.argtypes = [vp, u32, cu, cu, vp]; L.orc_deinterleave.restype = None
        L.orc_interleave.argtypes = [vp, u32, cu, cu, vp]; L.orc_interleave.restype = None
        L.orc_split_planes.argtypes = [vp, u32, cu, vp]; L.orc_split_planes.restype = None
        L.orc_merge_planes.argtypes = [vp, u32, cu, vp]; L.orc_merge_planes.restype = None
        L.orc_lz4_bound.argtypes = [u32]; L.orc_lz4_bound.restype = u32
        L.orc_lz4_compress.argtypes = [vp, u32, vp]; L.orc_lz4_compress.restype = ci
        L.orc_lz4_decompress.argtypes = [vp, u32, vp, u32]; L.orc_lz4_decompress.restype = ci
        L.orc_arch_new.restype = vp
        L.orc_arch_free.argtypes = [vp]; L.orc_arch_free.restype = None
        L.orc_arch_data.argtypes = [vp]; L.orc_arch_data.restype = vp
        L.orc_arch_size.argtypes = [vp]; L.orc_arch_size.restype = u64
        L.orc_arch_write_fp.argtypes = [vp, cu, u32, vp, u32, cu, ci]; L.orc_arch_write_fp.restype = None
        L.orc_arch_write_int.argtypes = [vp, cu, u32, vp, u32, cu]; L.orc_arch_write_int.restype = None
        _orc = L
    return _orc


# ---- numpy-level helpers over the restatement -------------------------------------------------

def fpc_encode(values, e1=None, e2=None):
    """values: 1-D float32/float64/uint32/uint64 array -> payload bytes"""
    a = np.ascontiguousarray(values)
    W = a.dtype.itemsize * 8
    if e1 is None:
        e1, e2 = (4, 10) if W == 32 else (20, 20)
    out = np.empty(orc().orc_fpc_bound(a.size, W), np.uint8)
    nb = orc().orc_fpc_encode(a.ctypes.data, a.size, W, e1, e2, out.ctypes.data)
    return out[:nb].tobytes()


def fpc_decode(payload, dtype):
    dtype = np.dtype(dtype)
    p = np.frombuffer(payload, np.uint8)
    if p.size < 5:
        return None
    n = int.from_bytes(bytes(p[1:5]), "big")
    out = np.empty(n, dtype)
    got = ctypes.c_uint32(0)
    ok = orc().orc_fpc_decode(p.ctypes.data, p.size, dtype.itemsize * 8, out.ctypes.data, n, ctypes.byref(got))
    return out if ok else None


def lz4_compress(data):
    a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
    out = np.empty(orc().orc_lz4_bound(a.size), np.uint8)
    nb = orc().orc_lz4_compress(a.ctypes.data if a.size else None, a.size, out.ctypes.data)
    return out[:nb].tobytes()


def lz4_decompress(block, size):
    p = np.frombuffer(block, np.uint8)
    out = np.empty(size, np.uint8)
    r = orc().orc_lz4_decompress(p.ctypes.data, p.size, out.ctypes.data, size)
    return out[:r].tobytes() if r >= 0 else None


def split_planes(arr):
    a = np.ascontiguousarray(arr)
    w = a.dtype.itemsize
    out = np.empty(a.size * w, np.uint8)
    orc().orc_split_planes(a.ctypes.data, a.size, w, out.ctypes.data)
    return out.reshape(w, a.size)


# stream tag / count / arity rules exactly as the reference writers apply them (trico.c:215-858)
_FP = {  # name: (tag, arity, dtype, count_scale)
    "vertices": (1, 3, np.float32, 1), "vertices_double": (2, 3, np.float64, 1),
    "uv_per_vertex": (5, 2, np.float32, 1), "uv_per_vertex_double": (5, 2, np.float64, 1),
    "uv_per_triangle": (7, 2, np.float32, 3), "uv_per_triangle_double": (7, 2, np.float64, 1),
    "vertex_normals": (9, 3, np.float32, 1), "vertex_normals_double": (10, 3, np.float64, 1),
    "triangle_normals": (11, 3, np.float32, 1), "triangle_normals_double": (12, 3, np.float64, 1),
    "attributes_float": (15, 1, np.float32, 1), "attributes_double": (16, 1, np.float64, 1),
}
_INT = {  # name: (tag, dtype, elems_per_count)
    "triangles": (3, np.uint32, 3), "triangles_long": (4, np.uint64, 3),
    "vertex_colors": (13, np.uint32, 1), "triangle_colors": (14, np.uint32, 1),
    "attributes_uint8": (17, np.uint8, 1), "attributes_uint16": (18, np.uint16, 1),
    "attributes_uint32": (19, np.uint32, 1), "attributes_uint64": (20, np.uint64, 1),
}
STREAM_KINDS = dict(_FP)
STREAM_KINDS.update(_INT)


class OracleArchive:
    """Archive writer on the restatement; write(name, data, count) mirrors trico_write_<name>."""

    def __init__(self):
        self.h = orc().orc_arch_new()

    def write(self, name, data, count):
        a = np.ascontiguousarray(data)
        if name in _FP:
            tag, arity, dt, scale = _FP[name]
            assert a.dtype == dt
            n = count * scale
            orc().orc_arch_write_fp(self.h, tag, n, a.ctypes.data, n, arity, dt().itemsize * 8)
        else:
            tag, dt, per = _INT[name]
            assert a.dtype == dt
            orc().orc_arch_write_int(self.h, tag, count, a.ctypes.data, count * per, dt().itemsize)
        return 1

    def tobytes(self):
        return ctypes.string_at(orc().orc_arch_data(self.h), orc().orc_arch_size(self.h))

    def close(self):
        if self.h:
            orc().orc_arch_free(self.h)
            self.h = None


# ---- the compiled reference ---------------------------------------------------------------------

def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The real reference library (trico_* API + trico_compress* + LZ4_*), or None if absent."""
    global _ref
    if _ref is None:
        if not have_ref():
            return None
        L = ctypes.CDLL(REF_SO)
        vp, u32, u64, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
        L.trico_open_archive_for_writing.restype = vp
        L.trico_open_archive_for_writing.argtypes = [u64]
        L.trico_open_archive_for_reading.restype = vp
        L.trico_open_archive_for_reading.argtypes = [vp, u64]
        L.trico_close_archive.argtypes = [vp]
        L.trico_close_archive.restype = None
        L.trico_get_buffer_pointer.restype = vp
        L.trico_get_buffer_pointer.argtypes = [vp]
        L.trico_get_size.restype = u64
        L.trico_get_size.argtypes = [vp]
        L.trico_get_next_stream_type.argtypes = [vp]
        L.trico_get_next_stream_type.restype = ci
        for name in STREAM_KINDS:
            w = getattr(L, "trico_write_" + name)
            w.argtypes = [vp, vp, u32]
            w.restype = ci
            r = getattr(L, "trico_read_" + name)
            r.argtypes = [vp, ctypes.POINTER(vp)]
            r.restype = ci
        for p in ("vertices", "triangles", "uvs", "normals", "colors", "attributes"):
            f = getattr(L, "trico_get_number_of_" + p)
            f.argtypes = [vp]
            f.restype = u32
        L.trico_compress.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(vp), vp, u32, u32, u32]
        L.trico_compress.restype = None
        L.trico_compress_double_precision.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(vp), vp, u32, u64, u64]
        L.trico_compress_double_precision.restype = None
        L.trico_decompress.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(vp), vp]
        L.trico_decompress.restype = None
        L.trico_decompress_double_precision.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(vp), vp]
        L.trico_decompress_double_precision.restype = None
        L.LZ4_compress_default.argtypes = [vp, vp, ci, ci]
        L.LZ4_compress_default.restype = ci
        L.LZ4_decompress_safe.argtypes = [vp, vp, ci, ci]
        L.LZ4_decompress_safe.restype = ci
        L.LZ4_compressBound.argtypes = [ci]
        L.LZ4_compressBound.restype = ci
        _ref = L
    return _ref


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def ref_fpc_encode(values, e1=None, e2=None):
    a = np.ascontiguousarray(values)
    L = ref()
    nb = ctypes.c_uint32(0)
    out = ctypes.c_void_p(None)
    if e1 is None:
        e1, e2 = (4, 10) if a.dtype.itemsize == 4 else (20, 20)
    if a.dtype.itemsize == 4:
        L.trico_compress(ctypes.byref(nb), ctypes.byref(out), a.ctypes.data, a.size, ctypes.c_uint32(e1), ctypes.c_uint32(e2))
    else:
        L.trico_compress_double_precision(ctypes.byref(nb), ctypes.byref(out), a.ctypes.data, a.size, ctypes.c_uint64(e1), ctypes.c_uint64(e2))
    b = ctypes.string_at(out, nb.value)
    _libc.free(out)
    return b


def ref_lz4_compress(data):
    a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
    L = ref()
    cap = L.LZ4_compressBound(a.size)
    out = np.empty(cap, np.uint8)
    nb = L.LZ4_compress_default(a.ctypes.data, out.ctypes.data, a.size, cap)
    return out[:nb].tobytes()


class RefArchive:
    """Archive writer on the compiled reference, same interface as OracleArchive."""

    def __init__(self, initial=1 << 20):
        self.h = ref().trico_open_archive_for_writing(initial)

    def write(self, name, data, count):
        a = np.ascontiguousarray(data)
        return getattr(ref(), "trico_write_" + name)(self.h, a.ctypes.data, count)

    def tobytes(self):
        L = ref()
        return ctypes.string_at(L.trico_get_buffer_pointer(self.h), L.trico_get_size(self.h))

    def close(self):
        if self.h:
            ref().trico_close_archive(self.h)
            self.h = None
