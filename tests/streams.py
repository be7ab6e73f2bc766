"""Shared stream tables for the tests (order and count divisors of tests/golden/allstreams.trc)."""
import numpy as np

# (writer/reader name, elements-per-count, peek name)
ALL_ORDER = [
    ("vertices", 3, "vertices"), ("triangles", 3, "triangles"), ("vertices_double", 3, "vertices"),
    ("triangles_long", 3, "triangles"), ("uv_per_vertex", 2, "uvs"), ("uv_per_triangle", 6, "uvs"),
    ("vertex_normals", 3, "normals"), ("vertex_normals_double", 3, "normals"), ("triangle_normals", 3, "normals"),
    ("triangle_normals_double", 3, "normals"), ("vertex_colors", 1, "colors"), ("triangle_colors", 1, "colors"),
    ("attributes_float", 1, "attributes"), ("attributes_double", 1, "attributes"), ("attributes_uint8", 1, "attributes"),
    ("attributes_uint16", 1, "attributes"), ("attributes_uint32", 1, "attributes"), ("attributes_uint64", 1, "attributes"),
]

STREAM_TAG = {
    "vertices": 1, "vertices_double": 2, "triangles": 3, "triangles_long": 4, "uv_per_vertex": 5,
    "uv_per_vertex_double": 6, "uv_per_triangle": 7, "uv_per_triangle_double": 8, "vertex_normals": 9,
    "vertex_normals_double": 10, "triangle_normals": 11, "triangle_normals_double": 12, "vertex_colors": 13,
    "triangle_colors": 14, "attributes_float": 15, "attributes_double": 16, "attributes_uint8": 17,
    "attributes_uint16": 18, "attributes_uint32": 19, "attributes_uint64": 20,
}


def mesh_streams(kind, W, H, seed=None):
    from trico_amd import meshgen as M
    if kind == "grid":
        v, t = M.grid(W, H) if seed is None else M.grid(W, H, seed)
        return [("vertices", v, W * H), ("triangles", t, 2 * W * H)]
    if kind == "walk":
        v, t = M.walk(W, H)
        return [("vertices", v, W * H), ("triangles", t, 2 * W * H)]
    v, n, uv, t = M.multi(W, H)
    return [("vertices_double", v, W * H), ("vertex_normals_double", n, W * H), ("uv_per_vertex", uv, W * H),
            ("triangles_long", t, 2 * W * H)]


def stream_count(name, div, arr):
    """count argument for trico_write_<name>: per-triangle uv passes nr_of_triangles (x3 positions, x2 comps)"""
    return arr.size // div
