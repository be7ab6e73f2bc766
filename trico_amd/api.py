"""ctypes binding of libtrico.so — the C-ABI of include/trico/trico.h and trico_hip.h.

The names, argument meaning and 1/0 error convention are the reference's (trico/trico.h:36-94);
this module only adds the argtypes.  Data arguments accept numpy arrays (host), torch CUDA tensors
(device, via data_ptr) or raw integer addresses.  Nothing here computes: every call goes through
the shared library, and loading fails loudly if the library has not been built.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRICO_AMD_LIB: another build of the same library (tests/test_gpu_selfcheck.py loads libtrico_testhooks.so through it)
LIB_PATH = os.environ.get("TRICO_AMD_LIB") or os.path.join(_HERE, "lib", "libtrico.so")

_lib = None

# reference: trico/trico.h:11-34
STREAM_TYPES = [
    "trico_empty", "trico_vertex_float_stream", "trico_vertex_double_stream",
    "trico_triangle_uint32_stream", "trico_triangle_uint64_stream",
    "trico_uv_per_vertex_float_stream", "trico_uv_per_vertex_double_stream",
    "trico_uv_per_triangle_float_stream", "trico_uv_per_triangle_double_stream",
    "trico_vertex_normal_float_stream", "trico_vertex_normal_double_stream",
    "trico_triangle_normal_float_stream", "trico_triangle_normal_double_stream",
    "trico_vertex_color_stream", "trico_triangle_color_stream",
    "trico_attribute_float_stream", "trico_attribute_double_stream",
    "trico_attribute_uint8_stream", "trico_attribute_uint16_stream",
    "trico_attribute_uint32_stream", "trico_attribute_uint64_stream",
]
for _i, _n in enumerate(STREAM_TYPES):
    globals()[_n] = _i

WRITERS = [
    "trico_write_vertices", "trico_write_vertices_double", "trico_write_triangles", "trico_write_triangles_long",
    "trico_write_uv_per_vertex", "trico_write_uv_per_vertex_double", "trico_write_uv_per_triangle",
    "trico_write_uv_per_triangle_double", "trico_write_vertex_normals", "trico_write_vertex_normals_double",
    "trico_write_triangle_normals", "trico_write_triangle_normals_double", "trico_write_vertex_colors",
    "trico_write_triangle_colors", "trico_write_attributes_float", "trico_write_attributes_double",
    "trico_write_attributes_uint8", "trico_write_attributes_uint16", "trico_write_attributes_uint32",
    "trico_write_attributes_uint64",
]
READERS = [w.replace("trico_write_", "trico_read_") for w in WRITERS]
PEEKS = [
    "trico_get_number_of_vertices", "trico_get_number_of_triangles", "trico_get_number_of_uvs",
    "trico_get_number_of_normals", "trico_get_number_of_colors", "trico_get_number_of_attributes",
]
API_SYMBOLS = (
    ["trico_open_archive_for_writing", "trico_open_archive_for_reading", "trico_close_archive"]
    + WRITERS + ["trico_get_buffer_pointer", "trico_get_size", "trico_get_version", "trico_get_next_stream_type"]
    + PEEKS + READERS + ["trico_skip_next_stream"]
)
HIP_SYMBOLS = [
    "trico_hip_available", "trico_hip_last_error", "trico_hip_ctx_create", "trico_hip_ctx_destroy",
    "trico_hip_set_stream", "trico_hip_synchronize", "trico_hip_pointer_is_device", "trico_hip_device_alloc",
    "trico_hip_device_free", "trico_hip_copy", "trico_hip_fpc_encode", "trico_hip_fpc_decode",
    "trico_hip_int_encode", "trico_hip_int_decode", "trico_hip_fetch_payload", "trico_hip_fetch_payloads", "trico_hip_fpc_encode_place", "trico_hip_payload_device_pointer",
    "trico_hip_open_archive_for_writing_device", "trico_hip_profile_enable", "trico_hip_profile_reset",
    "trico_hip_profile_ms", "trico_hip_last_stats", "trico_hip_encode_stats", "trico_hip_fpc32_code_sweep",
    "trico_hip_decode_jobs", "trico_hip_decode_jobs_reserve", "trico_hip_list_streams", "trico_hip_read_archives",
    "trico_hip_walk_frames", "trico_hip_release_workspaces",
]


class DecodeJob(ctypes.Structure):
    """trico_hip_decode_job (include/trico/trico_hip.h): one stream of a batched decode."""
    _fields_ = [("is_int", ctypes.c_int32), ("arity", ctypes.c_int32), ("width", ctypes.c_int32), ("n", ctypes.c_uint32),
                ("payloads", ctypes.c_void_p * 8), ("sizes", ctypes.c_uint32 * 8), ("dst", ctypes.c_void_p),
                ("ok", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class StreamInfo(ctypes.Structure):
    """trico_hip_stream_info: what trico_hip_list_streams reports per stream."""
    _fields_ = [("type", ctypes.c_int32), ("is_int", ctypes.c_int32), ("arity", ctypes.c_int32), ("width", ctypes.c_int32),
                ("count", ctypes.c_uint32), ("n", ctypes.c_uint32), ("decoded_bytes", ctypes.c_uint64),
                ("payload_bytes", ctypes.c_uint64)]

KERNEL_IDS = {
    "fpc32_encode": 0, "fpc64_encode": 1, "fpc32_decode": 2, "fpc64_decode": 3,
    "planes_split": 4, "planes_merge": 5, "lz4_encode": 6, "lz4_decode": 7,
}


def lib():
    """Load libtrico.so (once).  Raises if it is missing: there is no Python/CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libtrico.so not built: run `python -m trico_amd.build` (needs hipcc). "
            "The trico hot path has no CPU fallback.")
    # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7.  Importing torch
    # first makes the dynamic linker resolve libtrico.so's DT_NEEDED against that already-loaded
    # copy (same soname), so torch tensors and libtrico share one runtime / one KFD connection.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p, u32, u64, ci =ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.trico_open_archive_for_writing.restype = vp
    L.trico_open_archive_for_writing.argtypes = [u64]
    L.trico_hip_open_archive_for_writing_device.restype = vp
    L.trico_hip_open_archive_for_writing_device.argtypes = [u64]
    L.trico_open_archive_for_reading.restype = vp
    L.trico_open_archive_for_reading.argtypes = [u8p, u64]
    L.trico_close_archive.restype = None
    L.trico_close_archive.argtypes = [vp]
    for w in WRITERS:
        f = getattr(L, w)
        f.restype = ci
        f.argtypes = [vp, vp, u32]
    for r in READERS:
        f = getattr(L, r)
        f.restype = ci
        f.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p)]
    L.trico_skip_next_stream.restype = ci
    L.trico_skip_next_stream.argtypes = [vp]
    L.trico_get_buffer_pointer.restype = vp
    L.trico_get_buffer_pointer.argtypes = [vp]
    L.trico_get_size.restype = u64
    L.trico_get_size.argtypes = [vp]
    L.trico_get_version.restype = u32
    L.trico_get_version.argtypes = [vp]
    L.trico_get_next_stream_type.restype = ci
    L.trico_get_next_stream_type.argtypes = [vp]
    for p in PEEKS:
        f = getattr(L, p)
        f.restype = u32
        f.argtypes = [vp]
    # shim
    L.trico_hip_available.restype = ci
    L.trico_hip_last_error.restype = ctypes.c_char_p
    L.trico_hip_ctx_create.restype = vp
    L.trico_hip_ctx_destroy.argtypes = [vp]
    L.trico_hip_ctx_destroy.restype = None
    L.trico_hip_set_stream.argtypes = [vp]
    L.trico_hip_set_stream.restype = None
    L.trico_hip_synchronize.restype = ci
    L.trico_hip_pointer_is_device.argtypes = [vp]
    L.trico_hip_pointer_is_device.restype = ci
    L.trico_hip_device_alloc.argtypes = [ctypes.c_size_t]
    L.trico_hip_device_alloc.restype = vp
    L.trico_hip_device_free.argtypes = [vp]
    L.trico_hip_device_free.restype = None
    L.trico_hip_copy.argtypes = [vp, vp, ctypes.c_size_t]
    L.trico_hip_copy.restype = ci
    L.trico_hip_fpc_encode.argtypes = [vp, vp, u32, ci, ci, ctypes.POINTER(u32)]
    L.trico_hip_fpc_encode.restype = ci
    L.trico_hip_fpc_decode.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u32), ci, ci, u32, vp]
    L.trico_hip_fpc_decode.restype = ci
    L.trico_hip_int_encode.argtypes = [vp, vp, u32, ci, ctypes.POINTER(u32)]
    L.trico_hip_int_encode.restype = ci
    L.trico_hip_int_decode.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u32), ci, u32, vp]
    L.trico_hip_int_decode.restype = ci
    L.trico_hip_fetch_payload.argtypes = [vp, ci, vp]
    L.trico_hip_fetch_payloads.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.trico_hip_fetch_payloads.restype = ctypes.c_int
    L.trico_hip_fpc_encode_place.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    L.trico_hip_fpc_encode_place.restype = ctypes.c_int
    L.trico_hip_fetch_payload.restype = ci
    L.trico_hip_payload_device_pointer.argtypes = [vp, ci]
    L.trico_hip_payload_device_pointer.restype = vp
    L.trico_hip_last_stats.argtypes = [ctypes.POINTER(u32)]
    L.trico_hip_last_stats.restype = None
    L.trico_hip_encode_stats.argtypes = [ctypes.POINTER(u32)]
    L.trico_hip_encode_stats.restype = None
    L.trico_hip_fpc32_code_sweep.argtypes = []
    L.trico_hip_fpc32_code_sweep.restype = ctypes.c_int
    L.trico_hip_profile_enable.argtypes = [ci]
    L.trico_hip_profile_enable.restype = None
    L.trico_hip_profile_reset.restype = None
    L.trico_hip_profile_ms.argtypes = [ci, ctypes.POINTER(u64)]
    L.trico_hip_profile_ms.restype = ctypes.c_double
    # stream-sharded encoding + RCCL exchange (dist.hip)
    L.trico_hip_fpc_encode_component.argtypes = [vp, vp, u32, ci, ci, ci, ctypes.POINTER(u32)]
    L.trico_hip_fpc_encode_component.restype = ci
    L.trico_hip_int_encode_plane.argtypes = [vp, vp, u32, ci, ci, ctypes.POINTER(u32)]
    L.trico_hip_int_encode_plane.restype = ci
    L.trico_hip_append_encoded_stream.argtypes = [vp, ci, u32, ci, ctypes.POINTER(vp), ctypes.POINTER(u32)]
    L.trico_hip_append_encoded_stream.restype = ci
    L.trico_hip_comm_unique_id.argtypes = [vp]
    L.trico_hip_comm_unique_id.restype = ci
    L.trico_hip_comm_create.argtypes = [vp, ci, ci]
    L.trico_hip_comm_create.restype = vp
    L.trico_hip_comm_destroy.argtypes = [vp]
    L.trico_hip_comm_destroy.restype = None
    L.trico_hip_comm_gather.argtypes = [vp, vp, u64, ci, vp, u64, ctypes.POINTER(u64)]
    L.trico_hip_comm_gather.restype = ci
    # batched decode (engine.hip) and whole-archive reads
    L.trico_hip_decode_jobs.argtypes = [ctypes.POINTER(DecodeJob), ci]
    L.trico_hip_decode_jobs.restype = ci
    L.trico_hip_decode_jobs_reserve.argtypes = [ctypes.POINTER(DecodeJob), ci]
    L.trico_hip_decode_jobs_reserve.restype = ci
    L.trico_hip_list_streams.argtypes = [vp, ctypes.POINTER(StreamInfo), ci]
    L.trico_hip_list_streams.restype = ci
    L.trico_hip_read_archives.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(ctypes.POINTER(vp)), ctypes.POINTER(ci)]
    L.trico_hip_read_archives.restype = ci
    L.trico_hip_release_workspaces.restype = None
    _lib = L
    return L


def make_jobs(specs):
    """specs: dicts with is_int, arity, width, n, payloads (list of (address-like, size)), dst -> ctypes array of DecodeJob.
    The caller keeps the payload / dst objects alive."""
    arr = (DecodeJob * len(specs))()
    for j, sp in zip(arr, specs):
        j.is_int, j.arity, j.width, j.n = int(sp["is_int"]), int(sp.get("arity", 1)), int(sp["width"]), int(sp["n"])
        for c, (pp, sz) in enumerate(sp["payloads"]):
            j.payloads[c] = ptr(pp)
            j.sizes[c] = sz
        j.dst = ptr(sp["dst"])
    return arr


def list_streams(archive, cap=64):
    """trico_hip_list_streams: the streams from the cursor of a read archive to its end (list of StreamInfo), None if broken."""
    buf = (StreamInfo * cap)()
    n = lib().trico_hip_list_streams(archive.h, buf, cap)
    if n < 0:
        return None
    return [buf[i] for i in range(min(n, cap))]


def read_archives(archives, dsts):
    """trico_hip_read_archives: dsts[a] = list of destinations (arrays / tensors / None) for the remaining streams of
    archives[a]; ONE batched decode for all of them.  Returns 1 if every stream asked for was decoded."""
    n = len(archives)
    hs = (ctypes.c_void_p * n)(*[a.h for a in archives])
    rows = [(ctypes.c_void_p * max(1, len(d)))(*[ptr(x) for x in d]) for d in dsts]
    table = (ctypes.POINTER(ctypes.c_void_p) * n)(*[ctypes.cast(r, ctypes.POINTER(ctypes.c_void_p)) for r in rows])
    counts = (ctypes.c_int * n)(*[len(d) for d in dsts])
    return lib().trico_hip_read_archives(hs, n, table, counts)


def ptr(x):
    """Address of a numpy array, torch tensor, ctypes buffer, bytes object or int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, bytes):
        return ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p).value      # the bytes object itself: the caller keeps it alive
    if isinstance(x, bytearray):
        return ctypes.addressof((ctypes.c_char * len(x)).from_buffer(x))    # in place, no temporary copy
    return ctypes.addressof(x)


def last_error():
    return lib().trico_hip_last_error().decode()


class Archive:
    """Thin object wrapper over one archive handle; method names are the C names minus `trico_`."""

    def __init__(self, handle, keepalive=None):
        if not handle:
            raise RuntimeError("trico archive could not be opened: " + last_error())
        self.h = handle
        self._keep = keepalive

    @classmethod
    def open_for_writing(cls, initial_buffer_size=1 << 20, device=False):
        L = lib()
        if device:
            return cls(L.trico_hip_open_archive_for_writing_device(initial_buffer_size))
        return cls(L.trico_open_archive_for_writing(initial_buffer_size))

    @classmethod
    def open_for_reading(cls, data, size=None):
        """data: bytes / numpy uint8 array (host) or a torch CUDA uint8 tensor / int address (device)."""
        L = lib()
        if isinstance(data, (bytes, bytearray)):
            data = np.frombuffer(bytes(data), dtype=np.uint8)
        if size is None:
            size = data.nbytes if isinstance(data, np.ndarray) else data.numel() * data.element_size()
        h = L.trico_open_archive_for_reading(ptr(data), size)
        if not h:
            return None
        return cls(h, keepalive=data)

    def close(self):
        if self.h:
            lib().trico_close_archive(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # write side
    def write(self, name, data, count):
        return getattr(lib(), "trico_write_" + name)(self.h, ptr(data), count)

    def append_encoded_stream(self, stream_type, count_field, payloads, sizes):
        """trico_hip_append_encoded_stream: payloads = addresses / tensors / arrays of already encoded units."""
        n = len(payloads)
        pp = (ctypes.c_void_p * n)(*[ptr(p) for p in payloads])
        ss = (ctypes.c_uint32 * n)(*sizes)
        return lib().trico_hip_append_encoded_stream(self.h, stream_type, count_field, n, pp, ss)

    def get_size(self):
        return lib().trico_get_size(self.h)

    def get_buffer_pointer(self):
        return lib().trico_get_buffer_pointer(self.h)

    def tobytes(self):
        """Copy of the archive bytes (works for host and device archives)."""
        L = lib()
        n = L.trico_get_size(self.h)
        p = L.trico_get_buffer_pointer(self.h)
        if L.trico_hip_pointer_is_device(p):
            buf = (ctypes.c_uint8 * n)()
            if not L.trico_hip_copy(ctypes.addressof(buf), p, n):
                raise RuntimeError(last_error())
            return bytes(buf)
        return ctypes.string_at(p, n)

    # read side
    def get_version(self):
        return lib().trico_get_version(self.h)

    def get_next_stream_type(self):
        return lib().trico_get_next_stream_type(self.h)

    def get_number_of(self, what):
        return getattr(lib(), "trico_get_number_of_" + what)(self.h)

    def read(self, name, out):
        """out: caller-allocated array/tensor, or None to decode-and-discard (NULL)."""
        f = getattr(lib(), "trico_read_" + name)
        if out is None:
            return f(self.h, None)
        box = ctypes.c_void_p(ptr(out))
        return f(self.h, ctypes.byref(box))

    def read_alloc(self, name, count, dtype):
        """trico_read_attributes_float/double: the library mallocs the output; returns a numpy copy."""
        f = getattr(lib(), "trico_read_" + name)
        box = ctypes.c_void_p(None)
        ok = f(self.h, ctypes.byref(box))
        if not ok:
            return None
        arr = np.ctypeslib.as_array(ctypes.cast(box, ctypes.POINTER(ctypes.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).copy()
        libc = ctypes.CDLL(None)
        libc.free.argtypes = [ctypes.c_void_p]
        libc.free(box)
        return arr.view(dtype)

    def skip_next_stream(self):
        return lib().trico_skip_next_stream(self.h)
