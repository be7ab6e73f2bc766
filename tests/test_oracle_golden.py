"""CPU tier: the oracle restatement (oracle/trico_oracle.c) against the committed golden vectors that
were generated from the compiled reference (oracle/gen_golden.py).  Restates the reference's own
round-trip tests (trico.tests/fps_compression.cpp:79-234, int_compression.cpp:34-73,
trico_compression.cpp:14-177) on top of byte-level goldens the reference itself lacks."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from streams import ALL_ORDER, mesh_streams


def test_kat_fp(kat):
    n = 0
    for k in kat:
        if k["kind"] != "fpc":
            continue
        a = np.frombuffer(bytes.fromhex(k["input_hex"]), dtype=k["dtype"])
        assert O.fpc_encode(a).hex() == k["payload_hex"], k["name"]
        d = O.fpc_decode(bytes.fromhex(k["payload_hex"]), a.dtype)
        assert d is not None and d.tobytes() == a.tobytes(), k["name"]
        n += 1
    assert n >= 20


def test_kat_survey_hex():
    # SURVEY.md §8 known-answer vectors, typed in by hand from the survey (independent of kat.json)
    assert O.fpc_encode(np.array([1, 2, 3], np.float32)).hex() == (
        "25" "00000003" "2493e4" "3f800000" "40000000" "400000" "0000000000")
    assert O.fpc_encode(np.ones(8, np.float32)).hex() == "25" "00000008" "00002c" "3f800000" "00"
    assert O.fpc_encode(np.array([1, 2, 3], np.float64)).hex() == (
        "aa" "00000003" "88" "3ff0000000000000" "4000000000000000" "1f" "08000000000000" "00")
    assert O.fpc_encode(np.ones(4, np.float64)).hex() == "aa" "00000004" "98" "3ff0000000000000" "00" "00"


def test_kat_lz4(kat):
    n = 0
    for k in kat:
        if k["kind"] != "lz4":
            continue
        a = np.frombuffer(bytes.fromhex(k["input_hex"]), dtype=np.uint8)
        assert O.lz4_compress(a).hex() == k["payload_hex"], k["name"]
        if a.size:
            assert O.lz4_decompress(bytes.fromhex(k["payload_hex"]), a.size) == a.tobytes(), k["name"]
        n += 1
    assert n >= 20


@pytest.mark.parametrize("kind", ["grid", "walk", "multi"])
def test_small_archives(kind, gold_dir, native_libs):
    want = open(os.path.join(gold_dir, "%s_16x8.trc" % kind), "rb").read()
    a = O.OracleArchive()
    for name, data, count in mesh_streams(kind, 16, 8):
        a.write(name, data, count)
    got = a.tobytes()
    a.close()
    assert got == want


def test_allstreams_archive(allstreams, gold_dir):
    want = open(os.path.join(gold_dir, "allstreams.trc"), "rb").read()
    a = O.OracleArchive()
    for name, div, _ in ALL_ORDER:
        arr = allstreams[name]
        a.write(name, arr, arr.size // div)
    got = a.tobytes()
    a.close()
    assert got == want


@pytest.mark.parametrize("kind", ["grid", "walk", "multi"])
def test_config1_hashes(kind, hashes, native_libs):
    # BASELINE config 1 (1M vertices + 2M triangles, CPU round trip) and its walk/multi siblings
    h = hashes["%s_1000x1000" % kind]
    a = O.OracleArchive()
    for name, data, count in mesh_streams(kind, 1000, 1000):
        a.write(name, data, count)
    got = a.tobytes()
    a.close()
    assert len(got) == h["size"]
    assert hashlib.sha256(got).hexdigest() == h["sha256"]


def test_fp_roundtrip_tails():
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        for n in list(range(1, 20)) + [63, 64, 65, 1000, 4097]:
            a = (np.cumsum(rng.integers(-9, 10, n)) * 0.03125).astype(dt)
            a[rng.integers(0, n)] = dt(rng.standard_normal())
            p = O.fpc_encode(a)
            d = O.fpc_decode(p, dt)
            assert d is not None and d.tobytes() == a.tobytes()
            # truncated payloads must be rejected, not over-read
            assert O.fpc_decode(p[: len(p) // 2], dt) is None or n == 0


def test_lz4_edges():
    rng = np.random.default_rng(5)
    for n in (0, 1, 12, 13, 65546, 65547, 65548, 200000):
        a = (np.arange(n) // 5 % 256).astype(np.uint8) ^ (rng.integers(0, 256, n) > 250).astype(np.uint8)
        c = O.lz4_compress(a)
        if n:
            assert O.lz4_decompress(c, n) == a.tobytes()
            assert O.lz4_decompress(c, n - 1) is None          # output too small
            assert O.lz4_decompress(c[:-1], n) is None or n < 2  # truncated block
        else:
            assert c == b"\x00"


def test_planes_roundtrip():
    rng = np.random.default_rng(9)
    for dt in (np.uint16, np.uint32, np.uint64):
        a = rng.integers(0, 2 ** (8 * dt().itemsize), 1001, dtype=np.uint64).astype(dt)
        p = O.split_planes(a)
        for k in range(dt().itemsize):
            assert np.array_equal(p[k], ((a >> (8 * k)) & 0xFF).astype(np.uint8))


BUNNY_SHA256 = "91eb3432634421fc2e7807998fef01557df6dc7e732f821316bfb289cd3d1766"     # 584,613 B


def test_bunny_real_mesh(gold_dir):
    """The reference's only fixture (trico.tests/data/StanfordBunny.stl) through the restated STL reader
    (iostl.c:70-195 order) and the oracle: size and sha256 of the archive the reference produces."""
    from stl import read_stl
    v, t, nt = read_stl(os.path.join(gold_dir, "StanfordBunny.stl"))
    assert (v.size // 3, nt) == (34834, 69451)
    a = O.OracleArchive()
    a.write("vertices", v, v.size // 3)
    a.write("triangles", t, nt)
    got = a.tobytes()
    a.close()
    assert len(got) == 584613
    assert hashlib.sha256(got).hexdigest() == BUNNY_SHA256
