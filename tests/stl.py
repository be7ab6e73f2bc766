"""Binary STL -> (vertices, triangles) with the reference's vertex de-duplication order, for tests.
Restates trico_io/iostl.c:141-195 (read) and 70-138 (trico_remove_duplicate_vertices): corners are sorted
lexicographically by (x, y, z) with float comparisons, equal corners merge, the unique sorted list is the
vertex array and every triangle corner is remapped to its group.  (The reference's quicksort is unstable,
but merged corners are bitwise equal in this fixture, so the result does not depend on the order.)"""
import numpy as np


def read_stl(path):
    raw = open(path, "rb").read()
    nt = int(np.frombuffer(raw, np.uint32, 1, 80)[0])
    rec = np.frombuffer(raw, np.uint8, nt * 50, 84).reshape(nt, 50)
    corners = rec[:, 12:48].copy().view(np.float32).reshape(nt * 3, 3)      # normal(12) v0 v1 v2 attr(2)
    order = np.lexsort((corners[:, 2], corners[:, 1], corners[:, 0]))
    s = corners[order]
    new_group = np.ones(len(s), bool)
    new_group[1:] = np.any(s[1:] != s[:-1], axis=1)
    gid = np.cumsum(new_group) - 1
    vertices = np.ascontiguousarray(s[new_group]).reshape(-1)
    triangles = np.empty(nt * 3, np.uint32)
    triangles[order] = gid.astype(np.uint32)
    return vertices, triangles, nt
