"""Synthetic meshes of SURVEY.md §8(d) (grid / multi / walk) via libtrico_meshgen.so.
Test and bench input only; integer-derived, bit-reproducible."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtrico_meshgen.so")
_lib = None

GRID_SEED = 0x12345678
WALK_SEED = 0x9E3779B9


def _l():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtrico_meshgen.so not built: run `python -m trico_amd.build`")
        _lib = ctypes.CDLL(LIB_PATH)
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        _lib.trico_gen_grid.argtypes = [u32, u32, u32, vp, vp]
        _lib.trico_gen_walk.argtypes = [u32, u32, u32, vp, vp]
        _lib.trico_gen_multi.argtypes = [u32, u32, u32, vp, vp, vp, vp]
        for f in (_lib.trico_gen_grid, _lib.trico_gen_walk, _lib.trico_gen_multi):
            f.restype = None
    return _lib


def grid(W, H, seed=GRID_SEED, triangles=True):
    """float32 xyz vertices [W*H*3], uint32 triangle indices [2*W*H*3]"""
    v = np.empty(W * H * 3, np.float32)
    t = np.empty(W * H * 6, np.uint32) if triangles else None
    _l().trico_gen_grid(W, H, seed, v.ctypes.data, t.ctypes.data if triangles else None)
    return v, t


def walk(W, H, seed=WALK_SEED, triangles=True):
    v = np.empty(W * H * 3, np.float32)
    t = np.empty(W * H * 6, np.uint32) if triangles else None
    _l().trico_gen_walk(W, H, seed, v.ctypes.data, t.ctypes.data if triangles else None)
    return v, t


def multi(W, H, seed=GRID_SEED, triangles=True):
    """float64 vertices, float64 normals, float32 uv, uint64 triangles"""
    v = np.empty(W * H * 3, np.float64)
    n = np.empty(W * H * 3, np.float64)
    uv = np.empty(W * H * 2, np.float32)
    t = np.empty(W * H * 6, np.uint64) if triangles else None
    _l().trico_gen_multi(W, H, seed, v.ctypes.data, n.ctypes.data, uv.ctypes.data, t.ctypes.data if triangles else None)
    return v, n, uv, t
