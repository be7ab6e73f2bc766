"""K readers of one config-2 archive decoded as ONE batch (trico_hip_read_archives): seconds, GB/s of decoded bytes, repeats.

    python tools/bench_batch_decode.py [--mesh grid|walk] [--K 1,8,32] [--W 10000 --H 5000] [--no-reserve]

Prints one JSON line.  The environment (GPU_MAX_HW_QUEUES, TRICO_FPC32_CHAINS_PER_CU ...) is whatever the caller set."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trico_amd import api, meshgen


def batch_rows(Ks, d_v, d_t, nv, nt, raw_bytes, reserve=True, passes=2, profile=False):
    """One archive of (d_v, d_t) in HBM, K readers of it decoded as one batch.  Returns a list of dicts."""
    L = api.lib()
    a = api.Archive.open_for_writing(raw_bytes // 4, device=True)
    assert a.write("vertices", d_v, nv) == 1 and a.write("triangles", d_t, nt) == 1, api.last_error()
    torch.cuda.synchronize()
    rows = []
    stats = (ctypes.c_uint32 * 4)()
    for K in Ks:
        outs = [(torch.empty_like(d_v), torch.empty_like(d_t)) for _ in range(K)]
        for p in range(passes):
            readers = [api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size()) for _ in range(K)]
            L.trico_hip_last_stats(stats)
            rep0 = stats[2]
            if profile:
                L.trico_hip_profile_enable(1)
                L.trico_hip_profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ok = api.read_archives(readers, [[o[0], o[1]] for o in outs])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert ok == 1, api.last_error()
            for r in readers:
                r.close()
            for o in outs:
                assert torch.equal(o[0].view(torch.int32), d_v.view(torch.int32)) and torch.equal(o[1], d_t)
                o[0].zero_()
                o[1].zero_()
            L.trico_hip_last_stats(stats)
            row = {"archives": K, "float_chains": 3 * K, "seconds": round(dt, 3), "decode_GBps": round(K * raw_bytes / dt / 1e9, 3),
                   "repeats": int(stats[2] - rep0), "workspaces": "first use of this shape" if p == 0 else "kept from the pass before"}
            if profile:
                spans = ctypes.c_uint64(0)
                row["kernel_ms"] = {name: round(L.trico_hip_profile_ms(kid, ctypes.byref(spans)), 2) for name, kid in api.KERNEL_IDS.items()
                                    if name in ("fpc32_decode", "lz4_decode", "planes_merge")}
                L.trico_hip_profile_enable(0)
            rows.append(row)
        del outs
        torch.cuda.empty_cache()
    a.close()
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="grid")
    ap.add_argument("--K", default="1,8,32")
    ap.add_argument("--W", type=int, default=10000)
    ap.add_argument("--H", type=int, default=5000)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    v, t = (meshgen.grid if args.mesh == "grid" else meshgen.walk)(args.W, args.H)
    d_v = torch.from_numpy(v).cuda()
    d_t = torch.from_numpy(t.view(np.int32)).cuda()
    rows = batch_rows([int(k) for k in args.K.split(",")], d_v, d_t, args.W * args.H, 2 * args.W * args.H, v.nbytes + t.nbytes,
                      passes=args.passes, profile=args.profile)
    print(json.dumps({"mesh": "%s(%d,%d)" % (args.mesh, args.W, args.H), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                      "TRICO_FPC32_CHAINS_PER_CU": os.environ.get("TRICO_FPC32_CHAINS_PER_CU"), "rows": rows}))
