"""Stress: K readers decode full-size archives (grid / walk, optionally multi) at once, repeatedly; every decoded stream is compared with
its input.  Prints the failures with reader, round and stream.   python tools/stress_concurrent.py [rounds] [with_multi]"""
import os, sys, threading, time
# (no queue setting: the decode engine needs none)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from trico_amd import api
from streams import mesh_streams

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with_multi = len(sys.argv) > 2 and sys.argv[2] == "1"
W, H = 10000, 5000
kinds = ["grid", "walk", "multi"] if with_multi else ["grid", "walk"]
arch = {}
for k in kinds:
    dev = [(n, torch.from_numpy(a.view(np.uint8)).cuda(), c) for n, a, c in mesh_streams(k, W, H)]
    a = api.Archive.open_for_writing(1 << 20, device=True)
    for n, d, c in dev:
        assert a.write(n, d, c) == 1, api.last_error()
    arch[k] = (dev, a)
order = (["grid", "walk", "multi", "grid", "walk", "grid", "walk", "grid"] if with_multi else ["grid", "walk"] * 4)
fails = []


def reader(i, k, rnd):
    dev, a = arch[k]
    r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
    for n, d, c in dev:
        out = torch.empty_like(d)
        if r.read(n, out) != 1:
            fails.append((rnd, i, k, n, "read failed: " + api.last_error()))
            break
        if not torch.equal(out, d):
            bad = (out != d).nonzero()
            fails.append((rnd, i, k, n, "differs: %d bytes, first at %d" % (bad.numel(), int(bad[0]))))
        del out
    r.close()


for rnd in range(rounds):
    t0 = time.perf_counter()
    th = [threading.Thread(target=reader, args=(i, k, rnd)) for i, k in enumerate(order)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    print("round %d: %.2f s, failures so far: %d" % (rnd, time.perf_counter() - t0, len(fails)), flush=True)
for f in fails:
    print("FAIL", f)
import ctypes
st = (ctypes.c_uint32 * 4)()
api.lib().trico_hip_last_stats(st)
print("done: %d failures in %d rounds of %d readers; chain decodes repeated by the self-check: %d; payloads of another writer: %d"
      % (len(fails), rounds, len(order), st[2], st[3]))
