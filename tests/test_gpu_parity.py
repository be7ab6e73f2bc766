"""GPU tier (-m gpu): the HIP path, called through the C-ABI (libtrico.so), against the oracle and
the committed golden fixtures.  Bit-exact is the bar for everything here (integer/byte work)."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from streams import ALL_ORDER, STREAM_TAG, mesh_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(native_libs):
    L = native_libs.lib()
    assert L.trico_hip_available() == 1, "no HIP device: " + native_libs.last_error()
    return native_libs


def write_archive(api, streams, device=False, initial=1 << 20):
    a = api.Archive.open_for_writing(initial, device=device)
    for name, data, count in streams:
        assert a.write(name, data, count) == 1, (name, api.last_error())
    b = a.tobytes()
    a.close()
    return b


def oracle_archive(streams):
    a = O.OracleArchive()
    for name, data, count in streams:
        a.write(name, data, count)
    b = a.tobytes()
    a.close()
    return b


def read_back(api, blob, streams):
    r = api.Archive.open_for_reading(blob)
    assert r is not None
    for name, data, count in streams:
        assert r.get_next_stream_type() == STREAM_TAG[name]
        if name in ("attributes_float", "attributes_double"):
            got = r.read_alloc(name, count, data.dtype)
            assert got is not None, api.last_error()
        else:
            got = np.empty_like(data)
            assert r.read(name, got) == 1, (name, api.last_error())
        assert got.tobytes() == data.tobytes(), name
    assert r.get_next_stream_type() == api.trico_empty
    r.close()


@pytest.mark.parametrize("kind", ["grid", "walk", "multi"])
def test_small_golden_archives(api, gold_dir, kind):
    want = open(os.path.join(gold_dir, "%s_16x8.trc" % kind), "rb").read()
    streams = mesh_streams(kind, 16, 8)
    assert write_archive(api, streams) == want
    read_back(api, want, streams)


def test_allstreams_golden(api, gold_dir, allstreams):
    want = open(os.path.join(gold_dir, "allstreams.trc"), "rb").read()
    streams = [(name, allstreams[name], allstreams[name].size // div) for name, div, _ in ALL_ORDER]
    got = write_archive(api, streams, initial=16)      # tiny initial buffer: exercises growth
    assert got == want
    read_back(api, want, streams)


def test_kat_through_shim(api, kat):
    L = api.lib()
    ctx = L.trico_hip_ctx_create()
    assert ctx
    try:
        for k in kat:
            if k["n"] == 0:
                continue
            if k["kind"] == "fpc":
                a = np.frombuffer(bytes.fromhex(k["input_hex"]), dtype=k["dtype"]).copy()
                sizes = (ctypes.c_uint32 * 3)()
                assert L.trico_hip_fpc_encode(ctx, a.ctypes.data, a.size, 1, a.dtype.itemsize, sizes) == 1
            else:
                a = np.frombuffer(bytes.fromhex(k["input_hex"]), dtype=np.uint8).copy()
                sizes = (ctypes.c_uint32 * 8)()
                assert L.trico_hip_int_encode(ctx, a.ctypes.data, a.size, 1, sizes) == 1
            out = np.empty(sizes[0], np.uint8)
            assert L.trico_hip_fetch_payload(ctx, 0, out.ctypes.data) == 1
            assert out.tobytes().hex() == k["payload_hex"], k["name"]
    finally:
        L.trico_hip_ctx_destroy(ctx)


@pytest.mark.parametrize("n", list(range(1, 20)) + [63, 64, 65, 127, 128, 129, 1000, 4097])
def test_fp_tails_vs_oracle(api, n):
    rng = np.random.default_rng(100 + n)
    for dt, wname, rname in ((np.float32, "attributes_float", "vertices"), (np.float64, "attributes_double", "vertices_double")):
        a = (np.cumsum(rng.integers(-9, 10, 3 * n)) * 0.03125).astype(dt)
        a[rng.integers(0, 3 * n)] = dt(rng.standard_normal())
        streams = [(rname, a, n), (wname, a[:n].copy(), n)]
        got = write_archive(api, streams)
        assert got == oracle_archive(streams)
        read_back(api, got, streams)


@pytest.mark.parametrize("n", [1, 2, 3, 12, 13, 14, 15, 16, 17, 255, 4096, 65546, 65547, 65548, 100001])
def test_int_edges_vs_oracle(api, n):
    rng = np.random.default_rng(200 + n)
    a8 = ((np.arange(n) // 5 % 256).astype(np.uint8) ^ (rng.integers(0, 256, n) > 250).astype(np.uint8))
    a16 = (np.arange(n) % 1000).astype(np.uint16)
    a32 = rng.integers(0, max(2, n), n, dtype=np.uint32)
    a64 = np.arange(n, dtype=np.uint64) * 0x10001
    streams = [("attributes_uint8", a8, n), ("attributes_uint16", a16, n), ("attributes_uint32", a32, n),
               ("attributes_uint64", a64, n), ("vertex_colors", a32, n)]
    if n % 3 == 0:
        streams += [("triangles", a32, n // 3), ("triangles_long", a64, n // 3)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


@pytest.mark.parametrize("kind,W,H", [("grid", 200, 100), ("walk", 200, 100), ("multi", 100, 100)])
def test_meshes_vs_oracle(api, kind, W, H):
    streams = mesh_streams(kind, W, H)
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


def test_double_uv_writer_quirk(api):
    # trico.c:620-628: the double uv writers emit the FLOAT tags 5/7 and the unscaled count
    uv = (np.arange(40) / 64.0).astype(np.float64)
    streams = [("uv_per_vertex_double", uv, 20), ("uv_per_triangle_double", uv, 20)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    assert got[8] == 5
    r = api.Archive.open_for_reading(got)
    out = np.empty(40, np.float32)
    assert r.read("uv_per_vertex_double", np.empty(40, np.float64)) == 0   # tag is 5, not 6
    # The float reader accepts the tag.  The payload announces (20,20) tables, which are legal for a float stream too
    # (fpsc.c:214-217), so it is decoded as one - garbage or a clean failure, like the reference, but never a fault.
    assert r.read("uv_per_vertex", out) in (0, 1)
    r.close()


def test_device_pointers_and_device_archive(api):
    torch = pytest.importorskip("torch")
    streams = mesh_streams("grid", 64, 32)
    want = oracle_archive(streams)
    dev = [(name, torch.from_numpy(data.view(np.uint8).copy()).cuda(), count) for name, data, count in streams]
    a = api.Archive.open_for_writing(64, device=True)
    for name, t, count in dev:
        assert a.write(name, t, count) == 1, api.last_error()
    assert api.lib().trico_hip_pointer_is_device(a.get_buffer_pointer()) == 1
    assert a.tobytes() == want
    # decode straight from the device-resident archive into device outputs
    size = a.get_size()
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), size)
    assert r is not None
    for (name, data, count), (_, t, _) in zip(streams, dev):
        out = torch.zeros_like(t)
        assert r.read(name, out) == 1, api.last_error()
        assert torch.equal(out, t)
    r.close()
    a.close()


def test_corrupt_payload_is_rejected(api, gold_dir):
    blob = bytearray(open(os.path.join(gold_dir, "grid_16x8.trc"), "rb").read())
    # vertex stream: tag(1) count(4) nbytes(4) then payload: break the announced value count
    blob[8 + 1 + 4 + 4 + 4] ^= 0x55
    r = api.Archive.open_for_reading(bytes(blob))
    out = np.empty(16 * 8 * 3, np.float32)
    assert r.read("vertices", out) == 0
    assert r.get_next_stream_type() == api.trico_vertex_float_stream
    r.close()


def _fp32_data(kind, n, rng):
    if kind == "smooth":
        return (np.cumsum(rng.integers(-3, 4, n)) * 0.125).astype(np.float32)
    if kind == "noisy":
        return (np.cumsum(rng.integers(-128, 128, n)) / 1024.0).astype(np.float32)
    if kind == "randbits":       # every class/key collides all the time, all residual lengths
        return rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    if kind == "steps":          # long constant runs with jumps: code 0 runs + class changes
        return np.repeat(rng.standard_normal(n // 97 + 1).astype(np.float32), 97)[:n].copy()
    raise ValueError(kind)


@pytest.mark.parametrize("n", [1023, 1024, 1025, 2049, 4160, 65537, 250001, 1000003])
@pytest.mark.parametrize("kind", ["smooth", "noisy", "randbits", "steps"])
def test_fp32_segmented_encoder_vs_oracle(api, n, kind):
    """Segment/step boundaries of the throughput encoder (k_fpc32_encode.hip) for arity 1, 2 and 3."""
    rng = np.random.default_rng(n * 7 + len(kind))
    a3 = _fp32_data(kind, 3 * n, rng)
    streams = [("vertices", a3, n), ("uv_per_vertex", a3[: 2 * n].copy(), n), ("attributes_float", a3[:n].copy(), n)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    if n <= 70000:
        read_back(api, got, streams)


@pytest.mark.parametrize("kind", ["grid", "walk"])
def test_config1_vertices(api, kind, hashes):
    """BASELINE config 1 vertex stream (1M float vertices): payload sizes must match the reference's."""
    streams = mesh_streams(kind, 1000, 1000)[:1]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    want_sizes = hashes["%s_1000x1000" % kind]["streams"][0][2]
    import struct
    pos, sizes = 8 + 5, []
    for _ in range(3):
        nb = struct.unpack_from("<I", got, pos)[0]
        sizes.append(nb)
        pos += 4 + nb
    assert sizes == want_sizes


@pytest.mark.parametrize("kind", ["grid", "walk", "multi"])
def test_config1_full_archives_vs_reference_hash(api, kind, hashes):
    """BASELINE config 1 (1M vertices + 2M triangles) and its walk / multi siblings: whole archive sha256 and
    size against the golden produced by the compiled reference; then decode and compare with the input."""
    streams = mesh_streams(kind, 1000, 1000)
    got = write_archive(api, streams)
    h = hashes["%s_1000x1000" % kind]
    assert len(got) == h["size"]
    assert hashlib.sha256(got).hexdigest() == h["sha256"]
    read_back(api, got, streams)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 4097, 100001])
@pytest.mark.parametrize("kind", ["smooth", "noisy", "randbits", "steps"])
def test_fp64_wave_coder_vs_oracle(api, n, kind):
    """Wave-wide double coder (k_fpc64.hip): step/batch boundaries, odd tails, class collisions."""
    rng = np.random.default_rng(n * 11 + len(kind))
    if kind == "randbits":
        a3 = rng.integers(0, 2**63, 3 * n, dtype=np.uint64).view(np.float64)
    else:
        a3 = _fp32_data(kind, 3 * n, rng).astype(np.float64) + (1e-9 * np.arange(3 * n) if kind == "smooth" else 0.0)
    streams = [("vertices_double", a3, n), ("attributes_double", a3[:n].copy(), n)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


def test_bunny_real_mesh_gpu(api, gold_dir):
    """Real (non-synthetic) mesh: Stanford bunny -> archive must equal the reference's 584,613-byte golden."""
    from stl import read_stl
    v, t, nt = read_stl(os.path.join(gold_dir, "StanfordBunny.stl"))
    streams = [("vertices", v, v.size // 3), ("triangles", t, nt)]
    got = write_archive(api, streams)
    assert len(got) == 584613
    assert hashlib.sha256(got).hexdigest() == "91eb3432634421fc2e7807998fef01557df6dc7e732f821316bfb289cd3d1766"
    read_back(api, got, streams)
    # the same mesh widened to doubles / u64 indices (trico.tests/trico_compression.cpp:110-177) against the oracle
    streams = [("vertices_double", v.astype(np.float64), v.size // 3), ("triangles_long", t.astype(np.uint64), nt)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


# ---- read-ahead (archive.c start_readahead): streams decode concurrently, reads collect ----------------

def _allstreams_list(allstreams):
    return [(name, allstreams[name], allstreams[name].size // div) for name, div, _ in ALL_ORDER]


def test_readahead_with_skips(api, gold_dir, allstreams):
    """every other stream is skipped after the decode of all of them has been started"""
    blob = open(os.path.join(gold_dir, "allstreams.trc"), "rb").read()
    r = api.Archive.open_for_reading(blob)
    for i, (name, data, count) in enumerate(_allstreams_list(allstreams)):
        assert r.get_next_stream_type() == STREAM_TAG[name]
        if i % 2 == 1:
            assert r.skip_next_stream() == 1
            continue
        if name in ("attributes_float", "attributes_double"):
            got = r.read_alloc(name, count, data.dtype)
            assert got is not None, api.last_error()
        else:
            got = np.empty_like(data)
            assert r.read(name, got) == 1, (name, api.last_error())
        assert got.tobytes() == data.tobytes(), name
    assert r.get_next_stream_type() == api.trico_empty
    r.close()


def test_readahead_disabled_matches(api, gold_dir, allstreams, monkeypatch):
    monkeypatch.setenv("TRICO_HIP_READAHEAD_MB", "0")
    blob = open(os.path.join(gold_dir, "allstreams.trc"), "rb").read()
    read_back(api, blob, _allstreams_list(allstreams))


def test_readahead_corrupt_middle_stream(api, gold_dir):
    """grid_16x8.trc = vertices + triangles: a broken triangle plane fails in its own read, vertices still decode"""
    blob = bytearray(open(os.path.join(gold_dir, "grid_16x8.trc"), "rb").read())
    streams = mesh_streams("grid", 16, 8)
    r0 = api.Archive.open_for_reading(bytes(blob))
    v = np.empty_like(streams[0][1])
    assert r0.read("vertices", v) == 1
    r0.close()
    # locate the triangle stream: 8 (header) + tag + count + 3 x (nbytes, payload)
    pos = 8 + 1 + 4
    for _ in range(3):
        nb = int.from_bytes(blob[pos:pos + 4], "little")
        pos += 4 + nb
    assert blob[pos] == api.trico_triangle_uint32_stream
    nb0 = int.from_bytes(blob[pos + 5:pos + 9], "little")
    blob[pos + 9] = 0xF0            # first token of plane 0: 15+ literals announced ...
    for k in range(1, min(nb0, 6)):
        blob[pos + 9 + k] = 0xFF    # ... with a length that runs past the block
    r = api.Archive.open_for_reading(bytes(blob))
    got = np.empty_like(streams[0][1])
    assert r.read("vertices", got) == 1, api.last_error()
    assert got.tobytes() == streams[0][1].tobytes()
    tri = np.empty_like(streams[1][1])
    assert r.read("triangles", tri) == 0
    assert r.get_next_stream_type() == api.trico_triangle_uint32_stream
    r.close()


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_fp32_encoder_unaligned_device_pointer(api, shift):
    """the staged AoS loads use 16-byte accesses only when the vertex array is 16-byte aligned"""
    import torch
    rng = np.random.default_rng(77 + shift)
    n = 70001
    v = np.cumsum(rng.normal(0, 1e-3, (n, 3)), axis=0).astype(np.float32)
    flat = torch.empty(3 * n + 8, dtype=torch.float32, device="cuda")
    view = flat[shift:shift + 3 * n]
    view.copy_(torch.from_numpy(v.reshape(-1)))
    assert view.data_ptr() % 16 == 4 * shift
    a = api.Archive.open_for_writing(1 << 16)
    assert a.write("vertices", view, n) == 1, api.last_error()
    got = a.tobytes()
    a.close()
    assert got == oracle_archive([("vertices", v.reshape(-1), n)])


def test_mutated_archives_never_crash(api, gold_dir):
    """every reader either fails cleanly (0, cursor unchanged) or returns data; nothing may fault on corrupt input"""
    import random
    rnd = random.Random(99)
    names = ["grid_16x8.trc", "walk_16x8.trc", "multi_16x8.trc", "allstreams.trc"]
    reads = fails = 0
    for name in names:
        blob = open(os.path.join(gold_dir, name), "rb").read()
        for j in range(24):
            m = bytearray(blob)
            for _ in range(rnd.randrange(1, 5)):
                m[rnd.randrange(8, len(m))] = rnd.randrange(256)
            if j % 6 == 5:
                m = m[:rnd.randrange(16, len(m))]
            r = api.Archive.open_for_reading(bytes(m))
            assert r is not None
            for _ in range(40):
                st = r.get_next_stream_type()
                if st == api.trico_empty:
                    break
                name_of = [k for k, v in STREAM_TAG.items() if v == st]
                if not name_of or name_of[0] in ("attributes_float", "attributes_double"):
                    if r.skip_next_stream() != 1:
                        break
                    continue
                # generously sized destination: the (possibly corrupted) count field decides how much is written
                cnt = max(r.get_number_of(w) for w in ("vertices", "triangles", "uvs", "normals", "colors", "attributes"))
                if cnt > 1 << 16:
                    if r.skip_next_stream() != 1:
                        break
                    continue
                out = np.empty(cnt * 9 + 64, np.uint64)
                if r.read(name_of[0], out) == 1:
                    reads += 1
                else:
                    fails += 1
                    if r.skip_next_stream() != 1:
                        break
            r.close()
    assert reads > 0 and fails > 0


def test_concurrent_archives_from_threads(api, gold_dir):
    """one handle per thread (SURVEY 8(b) threading rule): encode + decode of different meshes at the same time"""
    import threading
    want = {k: open(os.path.join(gold_dir, "%s_16x8.trc" % k), "rb").read() for k in ("grid", "walk", "multi")}
    big = mesh_streams("walk", 300, 200)
    big_want = oracle_archive(big)
    errors = []

    def work(i):
        try:
            for rep in range(6):
                kind = ("grid", "walk", "multi")[(i + rep) % 3]
                streams = mesh_streams(kind, 16, 8)
                if write_archive(api, streams) != want[kind]:
                    errors.append((i, rep, kind, "encode"))
                read_back(api, want[kind], streams)
                if rep % 3 == 0:
                    if write_archive(api, big, device=(i % 2 == 0)) != big_want:
                        errors.append((i, rep, "big", "encode"))
                    read_back(api, big_want, big)
        except Exception as e:          # noqa: BLE001 - collected and asserted below
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.parametrize("kind", ["alphabet2", "alphabet4", "period_noise", "runs", "markov"])
@pytest.mark.parametrize("n", [70001, 1500000])
def test_lz4_short_sequence_regimes(api, kind, n):
    """byte streams that compress into very many short sequences with small offsets and overlapping matches: the
    regime of the batch-parsed LZ4 decoder path (k_lz4_decode.hip) and of the chunked compressor's stitch"""
    rng = np.random.default_rng(len(kind) * 1000 + n % 997)
    if kind == "alphabet2":
        data = rng.integers(0, 2, n, dtype=np.uint8) * 37
    elif kind == "alphabet4":
        data = rng.integers(0, 4, n, dtype=np.uint8)
    elif kind == "period_noise":
        p = np.tile(np.array([1, 2, 3, 1, 2, 7, 9], np.uint8), n // 7 + 1)[:n].copy()
        hit = rng.random(n) < 0.03
        p[hit] = rng.integers(0, 256, int(hit.sum()), dtype=np.uint8)
        data = p
    elif kind == "runs":
        lens = rng.integers(1, 40, n // 8 + 8)
        vals = rng.integers(0, 256, lens.size, dtype=np.uint8)
        data = np.repeat(vals, lens)[:n].astype(np.uint8)
    else:
        steps = rng.integers(-1, 2, n)
        data = (np.cumsum(steps) & 15).astype(np.uint8)
    data = np.ascontiguousarray(data)
    a = api.Archive.open_for_writing(1 << 16)
    assert a.write("attributes_uint8", data, n) == 1, api.last_error()
    got = a.tobytes()
    a.close()
    assert got == oracle_archive([("attributes_uint8", data, n)])
    r = api.Archive.open_for_reading(got)
    back = np.empty_like(data)
    assert r.read("attributes_uint8", back) == 1, api.last_error()
    r.close()
    assert back.tobytes() == data.tobytes()


@pytest.mark.parametrize("n", [262143, 262144, 262145, 300001, 700000, 1048576 + 7, 3999999])
@pytest.mark.parametrize("kind", ["markov", "period777", "zeros_tail"])
def test_lz4_planes_around_the_chunked_threshold(api, kind, n):
    """planes of 256 KiB and more take the chunk-speculative compressor (a single chunk at first, 64 KiB chunks on short sequences),
    smaller ones the one-workgroup compressor: the reference's bytes on both sides of the threshold"""
    rng = np.random.default_rng(n + len(kind))
    if kind == "markov":
        data = (np.cumsum(rng.integers(-1, 2, n)) & 15).astype(np.uint8)
    elif kind == "period777":
        data = np.tile(rng.integers(0, 256, 777, dtype=np.uint8), n // 777 + 1)[:n].copy()
        data[rng.integers(0, n, 25)] ^= 0x3c
    else:
        data = np.concatenate([rng.integers(0, 256, n // 3, dtype=np.uint8), np.zeros(n - n // 3, np.uint8)])
    data = np.ascontiguousarray(data)
    streams = [("attributes_uint8", data, n)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


@pytest.mark.parametrize("kind", ["period30k", "period_drift", "two_regimes", "random_then_zero", "short_sequences", "ramp_u32_plane",
                                  "long_runs", "many_long_runs", "open_match_ends", "long_period"])
def test_lz4_chunked_compressor_regimes(api, kind):
    """planes above the 4 MiB threshold of the chunk-speculative compressor (k_lz4_chunked.hip): long periodic matches whose
    speculative chunks get rejected (parallel alternative parses, adoption by the stitch walk), literal runs above 1 MiB (the
    big-run copy), short sequences (the small geometry chosen by the probe) - always the reference's bytes"""
    n = 9 * (1 << 20) + 12345
    rng = np.random.default_rng(4242 + len(kind))
    if kind == "period30k":
        base = rng.integers(0, 256, 30011, dtype=np.uint8)
        data = np.tile(base, n // base.size + 1)[:n].copy()
        hit = rng.integers(0, n, 40)
        data[hit] ^= 0x55
    elif kind == "period_drift":
        parts, pos = [], 0
        while pos < n:
            per = int(rng.integers(2000, 50000))
            reps = int(rng.integers(3, 40))
            parts.append(np.tile(rng.integers(0, 256, per, dtype=np.uint8), reps))
            pos += per * reps
        data = np.concatenate(parts)[:n].copy()
    elif kind == "two_regimes":
        a = np.tile(np.arange(256, dtype=np.uint8), n // 512 + 1)[:n // 2]
        steps = rng.integers(-1, 2, n - n // 2)
        b = (np.cumsum(steps) & 15).astype(np.uint8)
        data = np.concatenate([a, b])
    elif kind == "random_then_zero":
        data = np.concatenate([rng.integers(0, 256, 5 << 20, dtype=np.uint8), np.zeros(n - (5 << 20), np.uint8)])
    elif kind == "short_sequences":
        steps = rng.integers(-1, 2, n)
        data = (np.cumsum(steps) & 31).astype(np.uint8)
    elif kind == "long_runs":
        # constant runs of 0.3 .. 3.5 MiB: matches longer than a chunk are left open by the parse and counted by k_lz4_extend, several
        # per plane, some ending inside the 1 MiB a wave counts itself
        parts, pos = [], 0
        while pos < n:
            ln = int(rng.integers(300000, 3500000))
            parts.append(np.full(ln, int(rng.integers(0, 256)), np.uint8))
            pos += ln
        data = np.concatenate(parts)[:n].copy()
    elif kind == "many_long_runs":
        # fifty open matches in one plane: every workgroup of k_lz4_extend must hold the same list of them, in the same order
        parts = [np.full(int(rng.integers(1150000, 1700000)), v % 251, np.uint8) for v in range(50)]
        data = np.concatenate(parts)
        n = data.size
    elif kind == "open_match_ends":
        # periodic stretches that end 0, 1, 15, 16, 17 ... bytes behind the point where the parse leaves the match open, and around
        # the 64 KiB slices k_lz4_extend counts in
        parts = []
        for delta in (0, 1, 3, 4, 5, 15, 16, 17, 65535, 65536, 65537):
            per = rng.integers(0, 256, 777, dtype=np.uint8)
            ln = (1 << 20) + 777 + delta
            parts.append(np.tile(per, ln // 777 + 1)[:ln])
            parts.append(rng.integers(0, 256, 40, dtype=np.uint8))
        data = np.concatenate(parts)
        n = data.size
    elif kind == "long_period":
        # one period of 40,000 bytes for the whole plane: one open match with a large offset, running to the block's last bytes
        data = np.tile(rng.integers(0, 256, 40000, dtype=np.uint8), n // 40000 + 1)[:n].copy()
    else:
        data = (np.arange(n, dtype=np.uint64) * 3 // 7).astype(np.uint8)
    data = np.ascontiguousarray(data)
    a = api.Archive.open_for_writing(1 << 16)
    assert a.write("attributes_uint8", data, n) == 1, api.last_error()
    got = a.tobytes()
    a.close()
    assert got == oracle_archive([("attributes_uint8", data, n)])
    r = api.Archive.open_for_reading(got)
    back = np.empty_like(data)
    assert r.read("attributes_uint8", back) == 1, api.last_error()
    r.close()
    assert back.tobytes() == data.tobytes()


def test_u64_indices_with_empty_upper_planes(api):
    """triangles_long of a mesh with fewer than 2^24 vertices: five of the eight byte planes are zeros (one match per plane, left open by
    the parse of chunk 0 and counted by the whole device), the others as for u32 indices"""
    nt = 1_600_000
    rng = np.random.default_rng(31)
    base = np.arange(3 * nt, dtype=np.uint64) // 3
    t = (base + rng.integers(0, 50, 3 * nt).astype(np.uint64)) % np.uint64(900_000)
    streams = [("triangles_long", t, nt)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


@pytest.mark.parametrize("kind", ["ramp", "ramp_across_class", "ramp_wraps_zero", "big_stride", "constant", "constant_then_step",
                                  "ramp_restarts", "two_strides"])
def test_exact_hit_runs_of_the_float_decoder(api, kind):
    """bit patterns that make whole batches of exact FCM / DFCM hits (k_fpc32_decode.hip extrapolates such batches in closed form)
    and the cases where it must not: the common top four bits change inside a batch, the progression wraps, the stride is
    large, the run breaks or changes stride mid-batch.  The chain itself must get them right: no repeat by the self-check."""
    n = 64 * 41 + 17
    k = np.arange(n, dtype=np.uint64)
    if kind == "ramp":
        bits = 0x41200000 + 3 * k
    elif kind == "ramp_across_class":
        bits = 0x3FFFFC00 + 0x20 * k                       # crosses 0x40000000 inside the first batches
    elif kind == "ramp_wraps_zero":
        bits = (0x00000400 - 0x30 * k.astype(np.int64)).astype(np.uint64)      # runs below zero: wraps to 0xffffffxx
    elif kind == "big_stride":
        bits = 0x10000000 + 0x04000000 * k                 # stride 2^26: one class change every four values
    elif kind == "constant":
        bits = np.full(n, 0x42F6E979, np.uint64)
    elif kind == "constant_then_step":
        bits = np.where(k < 64 * 7 + 31, 0x42F6E979, 0x42F6E979 + 5 * (k - (64 * 7 + 31)))
    elif kind == "ramp_restarts":
        bits = 0x41200000 + 7 * (k % 1000)
    else:
        bits = np.where((k // 500) % 2 == 0, 0x41200000 + 3 * k, 0x41200000 + 3 * k + 11 * (k % 500))
    data = np.ascontiguousarray((np.asarray(bits, dtype=np.uint64) & 0xFFFFFFFF).astype(np.uint32).view(np.float32))
    a = api.Archive.open_for_writing(1 << 16)
    assert a.write("attributes_float", data, n) == 1, api.last_error()
    got = a.tobytes()
    a.close()
    assert got == oracle_archive([("attributes_float", data, n)])
    stats = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(stats)
    before = stats[2]
    r = api.Archive.open_for_reading(got)
    back = r.read_alloc("attributes_float", n, np.float32)
    assert back is not None, api.last_error()
    r.close()
    assert back.tobytes() == data.tobytes()
    api.lib().trico_hip_last_stats(stats)
    assert stats[2] == before                          # the decoder chain was right by itself
