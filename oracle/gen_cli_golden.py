"""Generates tests/golden/cli/: small STL / PLY inputs (synthetic, written by this script) and what the
REFERENCE's readers and command line tools make of them (oracle/_ref, built by oracle/Makefile from the
sources under /root/reference).  Fixtures are data only: input files, arrays, output files, hashes.

    python oracle/gen_cli_golden.py
"""
import ctypes
import hashlib
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "cli")
REF = os.path.join(ROOT, "oracle", "_ref")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(20261003)


# ---- inputs ---------------------------------------------------------------------------------------------
def lattice(w, h):
    """w x h height field, two triangles per cell, with per-vertex normals / colours and per-face uv"""
    ys, xs = np.mgrid[0:h, 0:w]
    z = 0.25 * np.sin(xs * 0.7) * np.cos(ys * 0.4) + rng.normal(0, 0.01, xs.shape)
    v = np.stack([xs * 0.5, ys * 0.5, z], -1).reshape(-1, 3).astype(np.float32)
    n = rng.normal(0, 1, v.shape)
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    c = rng.integers(0, 256, (v.shape[0], 4), dtype=np.uint8)
    i00 = (ys[:-1, :-1] * w + xs[:-1, :-1]).ravel()
    t = np.concatenate([np.stack([i00, i00 + 1, i00 + w], -1), np.stack([i00 + 1, i00 + w + 1, i00 + w], -1)]).astype(np.uint32)
    uv = rng.random((t.shape[0], 6), dtype=np.float32)
    return v, n, c, t, uv


def write_stl(path, v, t, normals=None, attrs=None, dup_zero=False):
    with open(path, "wb") as f:
        f.write(b"binary stl fixture".ljust(80, b" "))
        f.write(struct.pack("<I", t.shape[0]))
        for i, tri in enumerate(t):
            nrm = normals[i] if normals is not None else (0.0, 0.0, 0.0)
            f.write(struct.pack("<3f", *nrm))
            for k in tri:
                p = v[k].copy()
                if dup_zero and (i + int(k)) % 3 == 0:
                    p = np.where(p == 0, np.float32(-0.0), p)        # +0.0 / -0.0 twins of the same position
                f.write(struct.pack("<3f", *p))
            f.write(struct.pack("<H", int(attrs[i]) if attrs is not None else 0))


def ply_header(fmt, nv, nt, vprops, fprops, crlf=False, extra=()):
    lines = ["ply", "format %s 1.0" % fmt, "comment trico cli fixture"] + list(extra)
    lines += ["element vertex %d" % nv] + ["property %s" % p for p in vprops]
    lines += ["element face %d" % nt] + ["property %s" % p for p in fprops]
    lines += ["end_header"]
    nl = "\r\n" if crlf else "\n"
    return (nl.join(lines) + nl).encode()


def make_inputs():
    v, n, c, t, uv = lattice(9, 7)
    nt = t.shape[0]
    tn = rng.normal(0, 1, (nt, 3)).astype(np.float32)
    at = rng.integers(0, 65536, nt).astype(np.uint16)
    write_stl(os.path.join(OUT, "lattice.stl"), v, t, tn, at)
    vz = v.copy(); vz[:, 2] = 0.0; vz[::4, 0] = 0.0
    write_stl(os.path.join(OUT, "zeros.stl"), vz, t, tn, at, dup_zero=True)
    # ascii: everything, texcoord lists of varying length, one quad face (only 3 indices are kept)
    with open(os.path.join(OUT, "full_ascii.ply"), "wb") as f:
        f.write(ply_header("ascii", len(v), nt,
                           ["float x", "float y", "float z", "float nx", "float ny", "float nz",
                            "uchar red", "uchar green", "uchar blue", "uchar alpha"],
                           ["list uchar int vertex_indices", "list uchar float texcoord"], extra=["obj_info made by gen_cli_golden.py"]))
        for i in range(len(v)):
            f.write(("%.9g %.9g %.9g %.9g %.9g %.9g %d %d %d %d\n" % (*v[i], *n[i], *c[i])).encode())
        for i in range(nt):
            idx = list(t[i]) + ([int(t[i][0])] if i == 5 else [])
            k = [6, 6, 4, 0, 8, 6][i % 6]
            vals = list(uv[i]) + [0.5, 0.25]
            f.write(("%d %s %d %s\n" % (len(idx), " ".join(str(int(a)) for a in idx), k, " ".join("%.9g" % a for a in vals[:k]))).rstrip().encode() + b"\n")
    # ascii with \r\n line ends, rgb without alpha under the short names, vertex_index spelling
    with open(os.path.join(OUT, "rgb_crlf.ply"), "wb") as f:
        f.write(ply_header("ascii", len(v), nt, ["float32 x", "float32 y", "float32 z", "uint8 r", "uint8 g", "uint8 b"],
                           ["list uint8 uint32 vertex_index"], crlf=True))
        for i in range(len(v)):
            f.write(("%.9g %.9g %.9g %d %d %d\r\n" % (*v[i], *c[i][:3])).encode())
        for i in range(nt):
            f.write(("3 %d %d %d\r\n" % tuple(t[i])).encode())
    # binary little endian: double coordinates (narrowed to float by the reader), an unrelated property and element
    with open(os.path.join(OUT, "double_le.ply"), "wb") as f:
        lines = ["ply", "format binary_little_endian 1.0", "element vertex %d" % len(v), "property double x", "property double y",
                 "property double z", "property short quality", "property float nx", "property float ny", "property float nz",
                 "element edge 2", "property int a", "property int b",
                 "element face %d" % nt, "property list ushort int vertex_indices", "end_header"]
        f.write(("\n".join(lines) + "\n").encode())
        vd = v.astype(np.float64) + 1e-9
        for i in range(len(v)):
            f.write(struct.pack("<3dh3f", *vd[i], i - 20, *n[i]))
        f.write(struct.pack("<4i", 0, 1, 1, 2))
        for i in range(nt):
            f.write(struct.pack("<H3i", 3, *[int(a) for a in t[i]]))
    # binary big endian with diffuse colours
    with open(os.path.join(OUT, "diffuse_be.ply"), "wb") as f:
        f.write(ply_header("binary_big_endian", len(v), nt,
                           ["float x", "float y", "float z", "uchar diffuse_red", "uchar diffuse_green", "uchar diffuse_blue"],
                           ["list uchar uint vertex_indices", "list uchar float texcoord"]))
        for i in range(len(v)):
            f.write(struct.pack(">3f3B", *v[i], *c[i][:3]))
        for i in range(nt):
            f.write(struct.pack(">B3I", 3, *[int(a) for a in t[i]]) + struct.pack(">B6f", 6, *uv[i]))
    # malformed: body one value short
    blob = open(os.path.join(OUT, "rgb_crlf.ply"), "rb").read()
    open(os.path.join(OUT, "truncated.ply"), "wb").write(blob[:-12])


# ---- the reference's readers ----------------------------------------------------------------------------
def ref_io():
    L = ctypes.CDLL(os.path.join(REF, "libtrico_io_ref.so"))
    return L


def take(ptr, n, ctype, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), (n,)).copy().view(dtype)


def read_stl_ref(L, path, full):
    nv, nt = ctypes.c_uint32(), ctypes.c_uint32()
    pv, pt, pn, pa = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    if full:
        rc = L.trico_read_stl_full(ctypes.byref(nv), ctypes.byref(pv), ctypes.byref(nt), ctypes.byref(pt), ctypes.byref(pn), ctypes.byref(pa), path.encode())
    else:
        rc = L.trico_read_stl(ctypes.byref(nv), ctypes.byref(pv), ctypes.byref(nt), ctypes.byref(pt), path.encode())
    out = {"rc": np.array(rc)}
    if rc == 1:
        out["vertices"] = take(pv.value, nv.value * 3, ctypes.c_float, np.float32)
        out["triangles"] = take(pt.value, nt.value * 3, ctypes.c_uint32, np.uint32)
        if full:
            out["normals"] = take(pn.value, nt.value * 3, ctypes.c_float, np.float32)
            out["attributes"] = take(pa.value, nt.value, ctypes.c_uint16, np.uint16)
    return out


def read_ply_ref(L, path):
    nv, nt = ctypes.c_uint32(), ctypes.c_uint32()
    p = [ctypes.c_void_p() for _ in range(5)]       # vertices, normals, colors, triangles, texcoords
    rc = L.trico_read_ply(ctypes.byref(nv), ctypes.byref(p[0]), ctypes.byref(p[1]), ctypes.byref(p[2]), ctypes.byref(nt),
                          ctypes.byref(p[3]), ctypes.byref(p[4]), path.encode())
    out = {"rc": np.array(rc)}
    if rc == 1:
        out["vertices"] = take(p[0].value, nv.value * 3, ctypes.c_float, np.float32)
        out["normals"] = take(p[1].value, nv.value * 3, ctypes.c_float, np.float32)
        out["colors"] = take(p[2].value, nv.value, ctypes.c_uint32, np.uint32)
        out["triangles"] = take(p[3].value, nt.value * 3, ctypes.c_uint32, np.uint32)
        out["texcoords"] = take(p[4].value, nt.value * 6, ctypes.c_float, np.float32)
        out["has"] = np.array([bool(x.value) for x in p])
    return out


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def run(exe, *args):
    r = subprocess.run([os.path.join(REF, exe)] + list(args), capture_output=True, text=True)
    return r.returncode, r.stdout


def main():
    make_inputs()
    L = ref_io()
    arrays = {}
    for name in ("lattice.stl", "zeros.stl"):
        for full in (0, 1):
            for k, a in read_stl_ref(L, os.path.join(OUT, name), full).items():
                arrays["%s/full%d/%s" % (name, full, k)] = a
    bunny = os.path.join(ROOT, "tests", "golden", "StanfordBunny.stl")
    for name in ("full_ascii.ply", "rgb_crlf.ply", "double_le.ply", "diffuse_be.ply", "truncated.ply"):
        for k, a in read_ply_ref(L, os.path.join(OUT, name)).items():
            arrays["%s/%s" % (name, k)] = a
    np.savez_compressed(os.path.join(OUT, "reader_arrays.npz"), **arrays)

    # the reference's writers, on the arrays its readers produced
    r = read_stl_ref(L, os.path.join(OUT, "lattice.stl"), 1)
    f32p, u32p, u16p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16)
    L.trico_write_stl(r["vertices"].ctypes.data_as(f32p), r["triangles"].ctypes.data_as(u32p), ctypes.c_uint32(len(r["triangles"]) // 3),
                      r["normals"].ctypes.data_as(f32p), r["attributes"].ctypes.data_as(u16p), os.path.join(OUT, "writer_full.stl").encode())
    L.trico_write_stl(r["vertices"].ctypes.data_as(f32p), r["triangles"].ctypes.data_as(u32p), ctypes.c_uint32(len(r["triangles"]) // 3),
                      None, None, os.path.join(OUT, "writer_bare.stl").encode())
    p = read_ply_ref(L, os.path.join(OUT, "full_ascii.ply"))
    L.trico_write_ply(ctypes.c_uint32(len(p["vertices"]) // 3), p["vertices"].ctypes.data_as(f32p), p["normals"].ctypes.data_as(f32p),
                      p["colors"].ctypes.data_as(u32p), ctypes.c_uint32(len(p["triangles"]) // 3), p["triangles"].ctypes.data_as(u32p),
                      p["texcoords"].ctypes.data_as(f32p), os.path.join(OUT, "writer_full.ply").encode())
    L.trico_write_ply(ctypes.c_uint32(len(p["vertices"]) // 3), p["vertices"].ctypes.data_as(f32p), None, None,
                      ctypes.c_uint32(len(p["triangles"]) // 3), p["triangles"].ctypes.data_as(u32p), None, os.path.join(OUT, "writer_bare.ply").encode())

    # the reference's command line tools: (name, encoder args, decoder output extension or None)
    cases = [
        ("lattice_plain", ["-i", "lattice.stl"], "stl"),
        ("lattice_normals_uint16", ["-i", "lattice.stl", "-plyskip", "normal", "-plyskip", "uint16"], "stl"),
        ("zeros_plain", ["-i", "zeros.stl"], "stl"),
        ("full_ascii_all", ["-i", "full_ascii.ply"], None),                        # uv stream: decoder output undefined in the reference
        ("full_ascii_nouv", ["-i", "full_ascii.ply", "-stladd", "tex_coord"], "ply"),
        ("full_ascii_bare", ["-i", "full_ascii.ply", "-stladd", "tex_coord", "-stladd", "normal", "-stladd", "color"], "stl"),
        ("rgb_crlf", ["-i", "rgb_crlf.ply"], "ply"),
        ("double_le", ["-i", "double_le.ply"], "ply"),
        ("diffuse_be_nouv", ["-i", "diffuse_be.ply", "-stladd", "tex_coord"], "stl"),
    ]
    manifest = {"cases": [], "bunny": {}, "errors": []}
    for name, eargs, dext in cases:
        trc = name + ".trc"
        rc, out = run("trico_encoder_ref", *[os.path.join(OUT, a) if a.endswith((".stl", ".ply")) else a for a in eargs], "-o", os.path.join(OUT, trc))
        assert rc == 0, (name, rc, out)
        entry = {"name": name, "encoder_args": eargs, "trc": trc, "trc_sha256": sha(os.path.join(OUT, trc))}
        if dext:
            dec = name + ".decoded." + dext
            rc, out = run("trico_decoder_ref", "-i", os.path.join(OUT, trc), "-o", os.path.join(OUT, dec))
            assert rc == 0, (name, rc, out)
            entry["decoded"] = dec
            # default output name / type when -o is absent
            tmp = os.path.join(OUT, "_tmp.trc")
            open(tmp, "wb").write(open(os.path.join(OUT, trc), "rb").read())
            run("trico_decoder_ref", "-i", tmp)
            made = [e for e in ("stl", "ply") if os.path.exists(os.path.join(OUT, "_tmp." + e))]
            entry["default_ext"] = made[0]
            entry["default_sha256"] = sha(os.path.join(OUT, "_tmp." + made[0]))
            for e in made:
                os.remove(os.path.join(OUT, "_tmp." + e))
            os.remove(tmp)
        manifest["cases"].append(entry)
    # bunny: hashes only (3.4 MB outputs)
    tmp_trc, tmp_stl = os.path.join(OUT, "_bunny.trc"), os.path.join(OUT, "_bunny.stl")
    run("trico_encoder_ref", "-i", bunny, "-o", tmp_trc)
    run("trico_decoder_ref", "-i", tmp_trc, "-o", tmp_stl)
    manifest["bunny"] = {"trc_sha256": sha(tmp_trc), "trc_size": os.path.getsize(tmp_trc), "decoded_stl_sha256": sha(tmp_stl)}
    os.remove(tmp_trc); os.remove(tmp_stl)
    # error behaviour: (args, exit code as the shell sees it, stdout)
    for args in (["-i"], ["-i", "x.obj"], ["-i", os.path.join(OUT, "truncated.ply")], ["-q", "x", "y"], ["-i", "a.stl", "-stladd", "bogus"],
                 ["-i", os.path.join(OUT, "missing.stl")], []):
        rc, out = run("trico_encoder_ref", *args)
        manifest["errors"].append({"tool": "trico_encoder", "args": [a.replace(OUT + os.sep, "") for a in args], "rc": rc,
                                   "stdout": out.replace(OUT + os.sep, "")})
    for args in (["-i"], ["-i", os.path.join(OUT, "lattice.stl")], ["-x", "a", "b"], ["-i", os.path.join(OUT, "missing.trc")], []):
        rc, out = run("trico_decoder_ref", *args)
        manifest["errors"].append({"tool": "trico_decoder", "args": [a.replace(OUT + os.sep, "") for a in args], "rc": rc,
                                   "stdout": out.replace(OUT + os.sep, "")})
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("wrote", OUT, "-", len(os.listdir(OUT)), "files,", sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
