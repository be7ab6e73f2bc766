"""The throughput encoder for doubles (trico_amd/csrc/hip/k_fpc64_sort.hip) against the oracle: the owners' walk (runs of equal
hashes, lanes that share a table entry inside one step, lists of very different lengths), and the sorting path that skewed
streams take instead.  Streams of 8,192 values and more take this encoder (TRICO_FPC64_SORT_MIN)."""
import numpy as np
import pytest

from test_gpu_parity import api, oracle_archive, read_back, write_archive   # noqa: F401  (api is a fixture)

pytestmark = pytest.mark.gpu


def _data(kind, n, rng):
    """n doubles"""
    if kind == "noisy":                       # every value its own hashes: one operation per value and table
        return rng.standard_normal(n) * 1000.0
    if kind == "randbits":                    # all 64 bits random: NaNs, infinities, denormals among them
        return rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64)
    if kind == "constant":                    # one run per table from the second value on
        return np.full(n, 1.0)
    if kind == "ramp":                        # a grid's x: constant stride, the FCM hash changes every few thousand values
        return np.arange(n, dtype=np.float64) * 0.25
    if kind == "two":                         # ABAB...: no runs, two FCM hashes: everything in two lists (the sorting path's case)
        return np.where(np.arange(n) % 2 == 0, 1.0, 1024.0)
    if kind == "four":                        # ABCD...: 16 lanes of every step meet at each of four entries
        return np.array([1.0, 3.0, 9.0, 27.0])[np.arange(n) % 4]
    if kind == "few":                         # 37 values in random order: entries shared by several lanes of a step, runs of length 1-3
        return (rng.integers(0, 37, n) * 17.0 + 0.5)
    if kind == "steps":                       # runs of random length 1..200 of one value each
        v = rng.standard_normal(n // 2 + 2) * 10.0
        return np.repeat(v, rng.integers(1, 200, v.size))[:n].copy()
    if kind == "quantised":                   # a normal's component: 1024 levels, noisy
        return (rng.integers(0, 1024, n) - 512) / 512.0
    if kind == "owner0":                      # noisy, but the two halves of every FCM hash are equal: one owner holds every FCM operation
        b = rng.integers(0, 2**64, n, dtype=np.uint64)
        return ((b & ~np.uint64(0x3ff << 44)) | (((b >> np.uint64(54)) & np.uint64(0x3ff)) << np.uint64(44))).view(np.float64)
    raise ValueError(kind)


KINDS = ["noisy", "randbits", "constant", "ramp", "two", "four", "few", "steps", "quantised", "owner0"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mode", ["default", "sort", "short_lists"])
def test_vec3_streams_vs_oracle(api, monkeypatch, kind, mode):
    if mode == "sort":
        monkeypatch.setenv("TRICO_FPC64_WALK_MAX", "0")          # always the sorting path
    elif mode == "short_lists":
        monkeypatch.setenv("TRICO_FPC64_WALK_MAX", "300")        # a table whose longest list has more than 300 operations sorts
    n = 70001
    rng = np.random.default_rng(len(kind) * 1000 + n)
    a3 = np.empty(3 * n)
    a3[0::3] = _data(kind, n, rng)
    a3[1::3] = _data("steps", n, rng)
    a3[2::3] = _data(kind, n, rng)[::-1]
    streams = [("vertices_double", a3, n), ("vertex_normals_double", a3[::-1].copy(), n)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)
    read_back(api, got, streams)


@pytest.mark.parametrize("n", [8192, 8193, 9001, 20000, 65536, 65537, 66001, 131072 + 63, 300000])
def test_scalar_and_uv_streams_vs_oracle(api, n):
    """arity 1 (attributes) and 2 (uv) through the same kernels; counts around the tile and step sizes"""
    rng = np.random.default_rng(n)
    att = _data("few", n, rng)
    att[n // 3: n // 2] = _data("noisy", n // 2 - n // 3, rng)
    uv = np.empty(2 * n)
    uv[0::2] = _data("ramp", n, rng)
    uv[1::2] = _data("quantised", n, rng)
    streams = [("attributes_double", att, n), ("uv_per_vertex_double", uv, n)]
    got = write_archive(api, streams)
    assert got == oracle_archive(streams)


def test_one_long_list_takes_the_sorting_path(api, monkeypatch):
    """A stream without runs whose FCM operations all belong to one owner: longer than the default limit (262,144), so the FCM table
    is sorted and the DFCM table walked; bytes as the oracle's either way."""
    n = 400000
    rng = np.random.default_rng(8)
    a3 = np.empty(3 * n)
    for c in range(3):
        a3[c::3] = _data("owner0", n, rng)
    streams = [("vertices_double", a3, n)]
    want = oracle_archive(streams)
    assert write_archive(api, streams) == want
    monkeypatch.setenv("TRICO_FPC64_WALK_MAX", "4000000")         # ... and walked all the same
    assert write_archive(api, streams) == want


def test_threads_write_double_archives_at_the_same_time(api):
    """One archive handle per thread (the reference's threading model), five threads: every thread's encoder has its own side stream,
    the host arrays (14 MB and more each) go through the one ring of pinned chunks in turns.  Bytes as the oracle's, three times over."""
    import threading
    kinds = ["noisy", "few", "steps", "quantised", "ramp"]
    sets = []
    for k, kind in enumerate(kinds):
        n = 200003 + 40000 * k
        rng = np.random.default_rng(77 + k)
        a3 = np.empty(3 * n)
        for c in range(3):
            a3[c::3] = _data(kind if c != 1 else "noisy", n, rng)
        streams = [("vertices_double", a3, n), ("vertex_normals_double", a3[::-1].copy(), n)]
        sets.append((streams, oracle_archive(streams)))
    gate = threading.Barrier(len(sets))
    bad = []

    def work(k):
        streams, want = sets[k]
        gate.wait()
        for _ in range(3):
            if write_archive(api, streams) != want:
                bad.append(k)

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(sets))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
