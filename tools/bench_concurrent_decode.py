"""BASELINE config 5 on one GPU: K .trc archives (config-2 meshes, different seeds) decoded concurrently, one host
thread per archive (ctypes releases the GIL) through the plain trico_read_* calls; the decode engine combines the calls that
arrive together into one batch (tools/bench_batch_decode.py hands the archives over as a batch explicitly).
Prints aggregate decode throughput (decoded bytes / wall time) for K = 1, 2, 4, 8 (TRICO_BENCH_READERS=1,8,16,32 for other
counts: readers beyond the eighth decode the archives of the first eight again, into buffers of their own).

    python tools/bench_concurrent_decode.py [W H]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen

W, H = (10000, 5000) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
KS = [int(x) for x in os.environ.get("TRICO_BENCH_READERS", "1,2,4,8").split(",")]
KMAX = min(8, max(KS))
L = api.lib()
dev = torch.device("cuda", 0)
nv, nt = W * H, 2 * W * H

archives, raws = [], []
for k in range(KMAX):
    v, t = meshgen.grid(W, H, meshgen.GRID_SEED + k)
    d_v, d_t = torch.from_numpy(v).to(dev), torch.from_numpy(t.view(np.int32)).to(dev)
    a = api.Archive.open_for_writing((v.nbytes + t.nbytes) // 4, device=True)
    assert a.write("vertices", d_v, nv) == 1 and a.write("triangles", d_t, nt) == 1
    archives.append(a)
    raws.append((d_v, d_t))
    print("encoded archive", k, a.get_size(), "bytes", flush=True)
outs = [(torch.empty_like(raws[k % KMAX][0]), torch.empty_like(raws[k % KMAX][1])) for k in range(max(KS))]
raw_bytes = nv * 12 + nt * 12


def decode(k, errs):
    r = api.Archive.open_for_reading(archives[k % KMAX].get_buffer_pointer(), archives[k % KMAX].get_size())
    ok = r.read("vertices", outs[k][0]) == 1 and r.read("triangles", outs[k][1]) == 1
    r.close()
    if not ok:
        errs.append(k)


results = []
for K in [k for k in KS for _ in range(2)]:      # every count twice: the second pass finds its device buffers in the library's pool
    errs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=decode, args=(k, errs)) for k in range(K)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not errs, errs
    bad = 0
    for k in range(K):
        for name, got, want in (("vertices", outs[k][0].view(torch.int32), raws[k % KMAX][0].view(torch.int32)),
                                ("triangles", outs[k][1], raws[k % KMAX][1])):
            if not torch.equal(got, want):
                idx = (got != want).nonzero()
                print("MISMATCH reader %d %s: %d of %d elements differ, first at %d, last at %d" % (
                    k, name, idx.shape[0], got.numel(), int(idx[0][0]), int(idx[-1][0])), flush=True)
                bad += 1
        outs[k][0].zero_()
        outs[k][1].zero_()
    print('pass with %d readers: %d bad streams' % (K, bad), flush=True)
    results.append({"archives": K, "seconds": round(dt, 3), "decode_GBps": round(K * raw_bytes / dt / 1e9, 3)})
    print(json.dumps(results[-1]), flush=True)
print(json.dumps({"workload": "grid(%d,%d) archives, decode only, one GPU" % (W, H), "results": results}))
