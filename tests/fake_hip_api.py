"""Stand-in for trico_amd.api on a host without a GPU: the same surface bench.py's rank body uses (Archive, lib(), last_error,
KERNEL_IDS, a unit encoder), on CPU tensors, with the oracle as the coder.  Test infrastructure only (tests/test_bench_ranks_host.py
passes it to `bench.py --backend gloo --api tests.fake_hip_api`): what it times means nothing, what it checks is that the code
every rank runs - sharding, exchange, framing, golden lookup, the JSON line - works with more than one rank."""
import ctypes
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from trico_amd.parallel import STREAM_SHAPES  # noqa: E402

KERNEL_IDS = {}
_DT = {4: (np.float32, np.uint32), 8: (np.float64, np.uint64)}


class _Lib:
    def trico_hip_available(self):
        return 1

    def trico_hip_profile_enable(self, on):
        return None

    def trico_hip_profile_reset(self):
        return None

    def trico_hip_profile_ms(self, kid, spans):
        return 0.0

    def trico_hip_fpc32_code_sweep(self):
        return 0


def lib():
    return _Lib()


def last_error():
    return "(stand-in api: no error text)"


def _np(t, dtype):
    return t.detach().cpu().contiguous().numpy().view(dtype)


class Archive:
    def __init__(self, data=b"Trco\x00\x00\x00\x00"):
        self.buf = bytearray(data)
        self.pos = 8
        self._pin = None

    @classmethod
    def open_for_writing(cls, initial_buffer_size=1 << 20, device=False):
        return cls()

    @classmethod
    def open_for_reading(cls, data, size=None):
        if isinstance(data, (bytes, bytearray)):
            return cls(bytes(data))
        return cls(ctypes.string_at(int(data), int(size)))

    def write(self, name, data, count):
        _, arity, width, _, _ = STREAM_SHAPES[name]
        a = O.OracleArchive()
        a.write(name, _np(data, _DT[width][0 if arity is not None else 1]), count)
        self.buf += a.tobytes()[8:]
        a.close()
        return 1

    def append_encoded_stream(self, stream_type, count_field, payloads, sizes):
        self.buf += struct.pack("<BI", stream_type, count_field)
        for p, n in zip(payloads, sizes):
            b = p.detach().cpu().numpy().tobytes()
            assert len(b) == n
            self.buf += struct.pack("<I", n) + b
        return 1

    def get_size(self):
        return len(self.buf)

    def get_buffer_pointer(self):
        self._pin = (ctypes.c_uint8 * len(self.buf)).from_buffer_copy(bytes(self.buf))
        return ctypes.addressof(self._pin)

    def tobytes(self):
        return bytes(self.buf)

    def read(self, name, out):
        tag, arity, width, mult, per = STREAM_SHAPES[name]
        b = self.buf
        if b[self.pos] != tag:
            return 0
        count = struct.unpack_from("<I", b, self.pos + 1)[0]
        self.pos += 5
        units = arity if arity is not None else width
        parts = []
        for _ in range(units):
            n = struct.unpack_from("<I", b, self.pos)[0]
            parts.append(bytes(b[self.pos + 4: self.pos + 4 + n]))
            self.pos += 4 + n
        if arity is not None:
            o = _np(out, _DT[width][0]).reshape(-1, arity)
            for c, p in enumerate(parts):
                comp = O.fpc_decode(p, _DT[width][0])
                if comp is None or comp.size != o.shape[0]:
                    return 0
                o[:, c] = comp
        else:
            o = _np(out, np.uint8).reshape(-1, width)
            for k, p in enumerate(parts):
                plane = O.lz4_decompress(p, o.shape[0])
                if plane is None or len(plane) != o.shape[0]:
                    return 0
                o[:, k] = np.frombuffer(plane, np.uint8)
        # (the tensors bench.py passes are CPU tensors here: the views above alias them)
        return 1

    def close(self):
        self.buf = bytearray()


def unit_encoder(api):
    """encode(name, data, count, unit) -> uint8 tensor with the payload of one component / byte plane (parallel.hip_unit_encoder)"""
    def encode(name, data, count, unit):
        _, arity, width, _, per = STREAM_SHAPES[name]
        if arity is not None:
            comp = np.ascontiguousarray(_np(data, _DT[width][0]).reshape(-1, arity)[:, unit])
            pay = O.fpc_encode(comp)
        else:
            plane = np.ascontiguousarray(_np(data, np.uint8).reshape(-1, width)[:, unit])
            pay = O.lz4_compress(plane)
        return torch.from_numpy(np.frombuffer(pay, np.uint8).copy())

    encode.close = lambda: None
    return encode
