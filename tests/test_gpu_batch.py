"""GPU tier: the batched decode (trico_hip_decode_jobs / trico_hip_read_archives, trico_amd/csrc/hip/engine.hip): every
chain of a batch in one kernel launch.  Results must be those of the one-by-one readers (trico.c:943-1668), bit for bit, for
host- and device-resident archives, and a malformed stream must fail alone."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from streams import ALL_ORDER, mesh_streams

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api(native_libs):
    L = native_libs.lib()
    assert L.trico_hip_available() == 1, "no HIP device: " + native_libs.last_error()
    return native_libs


def oracle_archive(streams):
    a = O.OracleArchive()
    for name, data, count in streams:
        a.write(name, data, count)
    b = a.tobytes()
    a.close()
    return b


def some_archives(allstreams):
    """(blob written by the ORACLE, its streams) for a mix of shapes: float chains of several lengths (tails of 0..63 values
    behind the 64-value batches), doubles, u32 / u64 planes, and the archive with all twenty stream types."""
    sets = [mesh_streams("grid", 256, 128), mesh_streams("walk", 100, 77), mesh_streams("multi", 64, 33),
            mesh_streams("grid", 16, 8), mesh_streams("grid", 193, 50, seed=0x12345679),
            [(name, allstreams[name], allstreams[name].size // div) for name, div, _ in ALL_ORDER]]
    return [(oracle_archive(s), s) for s in sets]


def empty_like_streams(streams):
    return [np.zeros_like(d) for _, d, _ in streams]


@pytest.mark.parametrize("where", ["host", "device"])
def test_read_archives_equals_one_by_one(api, allstreams, where):
    import torch
    sets = some_archives(allstreams)
    keep, archives, outs = [], [], []
    stats = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(stats)
    repeats_before = stats[2]
    for blob, streams in sets:
        if where == "device":
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            keep.append(t)
            r = api.Archive.open_for_reading(t)
        else:
            r = api.Archive.open_for_reading(blob)
        assert r is not None
        infos = api.list_streams(r)
        assert infos is not None and len(infos) == len(streams)
        for inf, (name, data, count) in zip(infos, streams):
            assert inf.decoded_bytes == data.nbytes, name
            assert inf.count == (count * 3 if name == "uv_per_triangle" else count), name
        archives.append(r)
        if where == "device":
            outs.append([torch.zeros(d.nbytes, dtype=torch.uint8, device="cuda") for _, d, _ in streams])
        else:
            outs.append(empty_like_streams(streams))
    assert api.read_archives(archives, outs) == 1, api.last_error()
    for r, (blob, streams), o in zip(archives, sets, outs):
        assert r.get_next_stream_type() == api.trico_empty
        for (name, data, _), got in zip(streams, o):
            g = got.cpu().numpy().tobytes() if where == "device" else got.tobytes()
            assert g == data.tobytes(), name
        r.close()
    api.lib().trico_hip_last_stats(stats)
    assert stats[2] == repeats_before, "a chain decode had to be repeated"


def test_release_workspaces_and_decode_again(api, allstreams):
    """trico_hip_release_workspaces gives the engine's buffers and the pool back; the next batch allocates again and decodes the same."""
    blob, streams = some_archives(allstreams)[0]
    for _ in range(2):
        r = api.Archive.open_for_reading(blob)
        outs = empty_like_streams(streams)
        assert api.read_archives([r], [outs]) == 1, api.last_error()
        assert all(o.tobytes() == d.tobytes() for o, (_, d, _) in zip(outs, streams))
        r.close()
        api.lib().trico_hip_release_workspaces()


def test_read_archives_prefix_and_skips(api, allstreams):
    """nstreams smaller than what is left: the rest stays unread; a NULL destination skips its stream."""
    blob, streams = some_archives(allstreams)[5]
    r = api.Archive.open_for_reading(blob)
    outs = empty_like_streams(streams)
    first = [outs[0], None, outs[2]]
    assert api.read_archives([r], [first]) == 1, api.last_error()
    assert outs[0].tobytes() == streams[0][1].tobytes() and outs[2].tobytes() == streams[2][1].tobytes()
    assert not outs[1].any()
    assert r.get_next_stream_type() == 4                       # triangles_long comes next, read the classic way
    assert r.read("triangles_long", outs[3]) == 1, api.last_error()
    assert outs[3].tobytes() == streams[3][1].tobytes()
    r.close()


def test_malformed_stream_fails_alone(api):
    """Two archives in one batch; the second archive's x stream is cut short inside its groups: only that stream fails, the
    cursor of its archive stays in front of it, everything else is decoded."""
    good = mesh_streams("grid", 128, 64)
    blob_a = oracle_archive(good)
    blob_b = bytearray(oracle_archive(mesh_streams("walk", 128, 64)))
    # framing: 8 header, type, count, then size of x at 13..16 and its payload; damage the value count inside the payload header
    blob_b[18] ^= 0x40
    ra = api.Archive.open_for_reading(blob_a)
    rb = api.Archive.open_for_reading(bytes(blob_b))
    oa, ob = empty_like_streams(good), empty_like_streams(good)
    assert api.read_archives([ra, rb], [oa, ob]) == 0
    assert ra.get_next_stream_type() == api.trico_empty
    assert oa[0].tobytes() == good[0][1].tobytes() and oa[1].tobytes() == good[1][1].tobytes()
    assert rb.get_next_stream_type() == 1                      # still in front of its vertices
    ra.close()
    rb.close()


def test_decode_jobs_direct(api):
    """trico_hip_decode_jobs on payloads taken out of an archive by hand: float + int job, host pointers."""
    import struct
    streams = mesh_streams("grid", 200, 100)
    blob = oracle_archive(streams)
    (v, t) = (streams[0][1], streams[1][1])
    pos = 8
    specs, keep = [], []
    for is_int, ncomp, width, n, ref in ((0, 3, 4, 200 * 100, v), (1, 4, 4, 3 * 2 * 200 * 100, t)):
        pos += 5
        pays = []
        for _ in range(ncomp):
            nb = struct.unpack_from("<I", blob, pos)[0]
            p = np.frombuffer(blob[pos + 4: pos + 4 + nb], np.uint8).copy()
            keep.append(p)
            pays.append((p, nb))
            pos += 4 + nb
        out = np.zeros_like(ref)
        keep.append(out)
        specs.append({"is_int": is_int, "arity": 3 if not is_int else 1, "width": width, "n": n, "payloads": pays, "dst": out})
    jobs = api.make_jobs(specs)
    assert api.lib().trico_hip_decode_jobs_reserve(jobs, len(specs)) == 1, api.last_error()
    assert api.lib().trico_hip_decode_jobs(jobs, len(specs)) == 1, api.last_error()
    assert all(j.ok == 1 for j in jobs)
    assert specs[0]["dst"].tobytes() == v.tobytes() and specs[1]["dst"].tobytes() == t.tobytes()


CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from oracle import oracle as O
from streams import mesh_streams
sets = [mesh_streams("grid", 256, 128), mesh_streams("walk", 100, 77), mesh_streams("grid", 64, 64), mesh_streams("grid", 193, 50),
        mesh_streams("walk", 64, 9)]
archives, outs = [], []
for s in sets:
    a = O.OracleArchive()
    for name, data, count in s:
        a.write(name, data, count)
    archives.append(api.Archive.open_for_reading(a.tobytes()))
    a.close()
    outs.append([np.zeros_like(d) for _, d, _ in s])
assert api.read_archives(archives, outs) == 1, api.last_error()
for s, o in zip(sets, outs):
    for (name, data, _), got in zip(s, o):
        assert got.tobytes() == data.tobytes(), name
print("BATCH OK")
"""


@pytest.mark.parametrize("cpw", [2, 4])
def test_several_chains_per_workgroup(cpw):
    """TRICO_FPC32_CHAINS_PER_CU: the same batch with two and four chains per workgroup (15 float chains: the last workgroup is
    partly empty)."""
    env = dict(os.environ)
    env["TRICO_FPC32_CHAINS_PER_CU"] = str(cpw)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "BATCH OK" in out.stdout, out.stdout + out.stderr


def test_product_library_has_no_sabotage_switch():
    env = dict(os.environ)
    env["TRICO_HIP_DECODE_SABOTAGE"] = "3"
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "BATCH OK" in out.stdout, out.stdout + out.stderr


def test_payload_of_another_writer_is_decoded_and_counted(api):
    """A float stream whose codes are decodable but not the ones the reference's encoder picks (value 2 of eight 1.0f: an exact
    FCM hit written as code 1 with a zero byte instead of code 0).  The self-check cannot code the values back to these bytes;
    after the repeat ladder the reference-order kernel's values stand, and word 3 of trico_hip_last_stats counts the stream."""
    import struct
    pay = bytes.fromhex("25" "00000008" "00006c" "3f800000" "00" "00")
    blob = struct.pack("<II", 0x6f637254, 0) + bytes([15]) + struct.pack("<I", 8) + struct.pack("<I", len(pay)) + pay
    stats = (ctypes.c_uint32 * 4)()
    api.lib().trico_hip_last_stats(stats)
    before = stats[3]
    L = api.lib()
    L.trico_hip_archive_other_writer_streams.restype = ctypes.c_uint32
    L.trico_hip_archive_other_writer_streams.argtypes = [ctypes.c_void_p]
    r = api.Archive.open_for_reading(blob)
    assert L.trico_hip_archive_other_writer_streams(r.h) == 0
    got = r.read_alloc("attributes_float", 8, np.float32)
    assert got is not None, api.last_error()
    assert got.tolist() == [1.0] * 8
    assert L.trico_hip_archive_other_writer_streams(r.h) == 1      # the per-handle query: no environment variable needed to see it
    r.close()
    api.lib().trico_hip_last_stats(stats)
    assert stats[3] == before + 1
    # a strict reader (trico_hip_set_strict, the programmatic TRICO_HIP_STRICT=1) refuses the stream instead
    L.trico_hip_set_strict(1)
    try:
        r = api.Archive.open_for_reading(blob)
        assert r.read_alloc("attributes_float", 8, np.float32) is None
        assert "strict" in api.last_error()
        r.close()
    finally:
        L.trico_hip_set_strict(-1)
    api.lib().trico_hip_last_stats(stats)
    before += 1
    # the canonical payload of the same values raises no flag
    pay = bytes.fromhex("25" "00000008" "00002c" "3f800000" "00")
    blob = struct.pack("<II", 0x6f637254, 0) + bytes([15]) + struct.pack("<I", 8) + struct.pack("<I", len(pay)) + pay
    r = api.Archive.open_for_reading(blob)
    got = r.read_alloc("attributes_float", 8, np.float32)
    assert got is not None and got.tolist() == [1.0] * 8
    r.close()
    api.lib().trico_hip_last_stats(stats)
    assert stats[3] == before + 1


def test_lds_only_decoder_as_first_choice():
    """TRICO_HIP_DECODE_ROBUST=1: float chains decoded by k_fpc32_decode_robust (tables, ring and counters in LDS, nothing behind the
    scalar cache) instead of the scalar chain - same values, through the batch engine and through the single-stream path."""
    env = dict(os.environ)
    env["TRICO_HIP_DECODE_ROBUST"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "BATCH OK" in out.stdout, out.stdout + out.stderr


ENC_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
from trico_amd import api
from oracle import oracle as O
from streams import mesh_streams
for kind, W, H in (("grid", 256, 128), ("walk", 300, 77), ("grid", 16, 8), ("walk", 1000, 1000)):
    s = mesh_streams(kind, W, H)
    a = api.Archive.open_for_writing(1 << 16)
    o = O.OracleArchive()
    for name, data, count in s:
        assert a.write(name, data, count) == 1, api.last_error()
        o.write(name, data, count)
    assert a.tobytes() == o.tobytes(), (kind, W, H)
    a.close(); o.close()
print("ENCODE OK")
"""


def test_the_device_passes_the_lane_order_test_of_the_exchange(api):
    """The float encoder resolves run starts with one ds_wrxchg_rtn_b32 per predictor and step (k_fpc32_sweep.hip), which rests on the
    LDS unit applying the lanes of one instruction in increasing lane order; the library tests that on the device before it uses the
    kernel (k_fpc32_xchg_selftest) and would fall back to the two-sweep coder with ballots otherwise.  On an MI355X the test has to
    pass, and the library has to choose the one-sweep coder by itself (no environment variable)."""
    if os.environ.get("TRICO_FPC32_XCHG") == "0" or os.environ.get("TRICO_FPC32_SWEEPS"):
        pytest.skip("coder chosen by the environment")
    assert api.lib().trico_hip_fpc32_code_sweep() == 3


@pytest.mark.parametrize("env_add", [{"TRICO_FPC32_SWEEPS": "2"}, {"TRICO_FPC32_XCHG": "0"}, {"TRICO_FPC32_ASM": "0"},
                                     {"TRICO_FPC32_REPLAY": "4"}, {"TRICO_FPC32_SPARE": "3", "TRICO_FPC32_REPLAY": "1"}])
def test_encoder_variants_write_the_same_bytes(env_add):
    """The other shapes of the float encoder: TRICO_FPC32_SWEEPS=2 = index sweep + code sweep with the exchange (round 3's encoder,
    kept for measurements); TRICO_FPC32_XCHG=0 = two sweeps with ballots, i.e. what a device that fails the lane-order test runs, what
    a stream is coded with again when the one-sweep coder raised a flag, and what the decoders' self-check re-encodes with;
    TRICO_FPC32_ASM=0 = the one-sweep coder with every step through the compiled code instead of the hand-written loop; _REPLAY = blocks
    of the segment before that a wave replays into its tables first; _SPARE = workgroup places per compute unit left free.  They decide
    time and traffic, never bytes.  The switches exist in the test-hooks build only."""
    env = dict(os.environ)
    env["TRICO_AMD_LIB"] = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")
    env.update(env_add)
    out = subprocess.run([sys.executable, "-c", ENC_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ENCODE OK" in out.stdout, out.stdout + out.stderr


def test_threads_with_one_archive_each_share_a_batch(api):
    """The reference's threading model: one archive handle per thread.  Six threads read six different archives at the same time
    through the plain trico_read_* calls; the engine combines what arrives together into one batch (engine.hip, "callers from
    several threads").  Every thread gets its own streams back, bit for bit."""
    import threading
    sets = [mesh_streams("grid", 256, 128), mesh_streams("walk", 100, 77), mesh_streams("multi", 64, 33),
            mesh_streams("grid", 193, 50, seed=0x12345679), mesh_streams("walk", 64, 9), mesh_streams("grid", 64, 64)]
    blobs = [oracle_archive(s) for s in sets]
    errs = []
    gate = threading.Barrier(len(sets))

    def work(k):
        try:
            r = api.Archive.open_for_reading(blobs[k])
            gate.wait()
            for name, data, count in sets[k]:
                got = np.zeros_like(data)
                if r.read(name, got) != 1:
                    errs.append((k, name, api.last_error()))
                    return
                if got.tobytes() != data.tobytes():
                    errs.append((k, name, "differs"))
                    return
            r.close()
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))

    for rep in range(3):
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(sets))]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert all(not t.is_alive() for t in th)
        assert not errs, errs


def test_batch_with_an_empty_stream_and_a_bad_job(api):
    """An empty float stream (count 0: header + pad group here, undefined in the reference) inside a batch goes through the
    single-stream path; a job with a nonsensical shape fails alone."""
    v = np.zeros(0, np.float32)
    t = np.arange(12, dtype=np.uint32)
    a = api.Archive.open_for_writing(64)
    assert a.write("vertices", v, 0) == 1 and a.write("triangles", t, 4) == 1, api.last_error()
    blob = a.tobytes()
    a.close()
    r = api.Archive.open_for_reading(blob)
    infos = api.list_streams(r)
    assert [i.n for i in infos] == [0, 12] and [i.decoded_bytes for i in infos] == [0, 48]
    out_v, out_t = np.zeros(1, np.float32), np.zeros(12, np.uint32)
    assert api.read_archives([r], [[out_v, out_t]]) == 1, api.last_error()
    assert out_t.tolist() == t.tolist() and r.get_next_stream_type() == api.trico_empty
    r.close()
    good = np.arange(64, dtype=np.uint8)
    jobs = api.make_jobs([{"is_int": 1, "width": 3, "n": 64, "payloads": [(good, 64)], "dst": np.zeros(64, np.uint8)},
                          {"is_int": 0, "arity": 7, "width": 4, "n": 64, "payloads": [(good, 64)], "dst": np.zeros(64, np.float32)}])
    assert api.lib().trico_hip_decode_jobs(jobs, 2) == 0
    assert jobs[0].ok == 0 and jobs[1].ok == 0
