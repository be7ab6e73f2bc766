"""Worker for test_gpu_lz4_chunked.py: run with small TRICO_LZ4_* chunk settings so that megabyte planes
take the chunk-speculative path (k_lz4_chunked.hip) with many chunks.  Prints one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from trico_amd import api
from oracle import oracle as O


def cases():
    rng = np.random.default_rng(42)
    n = 3 * 1000 * 1000 + 17
    yield "zeros", np.zeros(n, np.uint8)
    yield "random", rng.integers(0, 256, n, dtype=np.uint8)
    yield "period1536", np.tile((np.arange(1536) * 7 % 256).astype(np.uint8), n // 1536 + 1)[:n].copy()
    yield "runs", np.repeat(rng.integers(0, 256, n // 300 + 1, dtype=np.uint8), 300)[:n].copy()
    yield "semi", ((np.arange(n) // 5 % 256).astype(np.uint8) ^ (rng.integers(0, 256, n) > 250).astype(np.uint8))
    words = rng.integers(0, 256, (64, 4), dtype=np.uint8)          # immediate re-match chains: tiny dictionary of 4-byte words
    yield "dict4", words[rng.integers(0, 64, n // 4 + 1)].reshape(-1)[:n].copy()
    mix = np.concatenate([rng.integers(0, 256, 700000, dtype=np.uint8), np.zeros(900000, np.uint8),
                          np.tile(np.arange(251, dtype=np.uint8), 3000), rng.integers(0, 4, 800000, dtype=np.uint8)])
    yield "mixed", mix
    lowent = rng.integers(0, 3, n, dtype=np.uint8)
    yield "lowentropy", lowent


def main():
    ok = True
    for name, a in cases():
        w = api.Archive.open_for_writing(1 << 16)
        assert w.write("attributes_uint8", a, a.size) == 1, api.last_error()
        got = w.tobytes()
        w.close()
        import ctypes
        st = (ctypes.c_uint32 * 4)()
        api.lib().trico_hip_last_stats(st)
        o = O.OracleArchive()
        o.write("attributes_uint8", a, a.size)
        want = o.tobytes()
        o.close()
        same = got == want
        back = np.empty_like(a)
        r = api.Archive.open_for_reading(got)
        rd = r.read("attributes_uint8", back) == 1 and back.tobytes() == a.tobytes()
        r.close()
        print("%s n=%d archive=%d exact=%s roundtrip=%s chunks_accepted=%d reparsed=%d" % (name, a.size, len(got), same, rd, st[0], st[1]), flush=True)
        ok = ok and same and rd
    # u32 triangles through the same path (4 planes at once)
    t = (np.arange(3 * 400000, dtype=np.uint32) * 3) % 1000003
    w = api.Archive.open_for_writing(1 << 16)
    assert w.write("triangles", t, 400000) == 1, api.last_error()
    got = w.tobytes()
    w.close()
    o = O.OracleArchive()
    o.write("triangles", t, 400000)
    same = got == o.tobytes()
    o.close()
    print("triangles exact=%s" % same, flush=True)
    ok = ok and same
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
