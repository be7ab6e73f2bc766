"""bench.py — driver benchmark: Trico encode -> decode round trip on device-resident synthetic meshes.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one pass of the hot path over one mesh per GPU: raw arrays resident in HBM ->
trico_write_vertices + trico_write_triangles into a device-resident .trc archive -> (N > 1: RCCL gather
of the archives to rank 0 over xGMI) -> trico_open_archive_for_reading + trico_read_* back into HBM.
Workload = BASELINE.json configs[1]: grid(10000,5000) = 50M float xyz vertices + 100M uint32 triangles
(1.8e9 raw bytes) per GPU; with N GPUs every rank codes its own mesh (configs[3], seeds 0x12345678+rank),
so scaling is weak.  Before timing, the archive is checked against the reference's sha256
(tests/golden/hashes.json) and the decoded arrays against the input (bit-exact).

value = raw input bytes of all ranks / wall time of (encode + gather + decode), GB/s.
roofline  = the float-vertex encoder (north_star's target kernel): algorithmic bytes (raw vertex bytes
            + their payload bytes) / average device time of its launch sequence, vs 8 TB/s HBM3E.
cpu_baseline = the reference itself (oracle/_ref, built from /root/reference) when that library is
            present, else the oracle port, on the SAME mesh, on this box's host cores: 1 thread (the reference's
            execution model) and one thread per independent stream; `nproc` states the cores of the box.
roofline_all / decode_model: every other stage of the step against the same HBM peak (algorithmic bytes of SURVEY 8(d) /
            live hipEvent spans), and what bounds the chain decoders: nanoseconds and cycles per value of each component chain.
other_mesh / pcie_inclusive / decode_concurrent / config3: the same step on the walk mesh, through host pointers, BASELINE
            configs[4]'s shape on one GPU (8 / 16 / 32 / 64 archives decoded as ONE batch, trico_hip_read_archives, with the hardware
            queue count unset, 1, 4 and 32), and BASELINE configs[2] (double vertices + normals + float uv, + u64 triangles so that the
            archive is the reference's golden) - N = 1 only, outside the timed region.  --quick skips them.
Every block that is not the headline carries the reference's time beside it (`cpu_twin`, kind "reference", the sample stated):
config3 on a tenth of its mesh with 1 thread and one thread per independent stream, the walk mesh in full, config5_mixed on meshes
of a quarter of the vertices with one thread per stream of every archive.
`--mode decode-mixed` is BASELINE configs[4] as the N-GPU job: rank r builds archive kind[r mod 3] of (grid, walk, multi) on its GPU
(untimed) and decodes it K times; no collective; value = decoded bytes of all ranks / wall time.
`python bench.py --gpus N` starts its N ranks itself (torch.distributed.run, RCCL) when no launcher did.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (No GPU_MAX_HW_QUEUES here any more: round 2 needed 16 hardware queues because every stream decoded on a HIP stream of its own;
# the decode engine launches all chains of a batch as one grid on three HIP streams, whatever the process allows.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def torch_device_count():
    """Devices this process could use, without initialising any of them (a parent that has touched the GPU must not exec or fork
    ranks; torch.cuda.device_count() only counts on this image)."""
    import torch
    return int(torch.cuda.device_count())


def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` without a launcher: start N ranks (fresh processes, one per GPU, RCCL rendezvous on
    127.0.0.1) through torch.distributed.run BEFORE this process touches the GPU, relay their output and exit with their
    code.  Under a launcher (WORLD_SIZE set) this is a no-op."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    have = torch_device_count()
    if have < n:
        # (before any rendezvous: N ranks on fewer devices would sit in the RCCL bootstrap until somebody kills them)
        sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible: not started\n" % (n, have))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


if __name__ == "__main__":
    _spawn_ranks_if_needed()

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _pmc_file_traffic(path, family="k_fpc32", per_launch_of=("k_fpc32_sweep", "k_fpc32_code")):
    """bytes per launch sequence from a tools/pmc_summary.py text (KB per dispatch; FETCH_SIZE doubled per the gfx950 rule of
    MI355X_MICROARCH.md, validated for this kernel's reads in profiles/r02_ubench_fetch_size.txt).  `family`: prefix of the
    kernels that count (None: every kernel but the runtime's own fills and copies); `per_launch_of`: the kernel whose dispatches
    count the launch sequences (the first of them the file holds)."""
    import re
    mine = (lambda k: k.startswith(family)) if family else (lambda k: not k.startswith("__amd"))
    kernel, disp, fetch, write = None, {}, {}, {}
    section = "grid"                                   # (round 4's files hold a "## grid" and a "## walk" section; older ones the grid only)
    for line in open(path):
        if line.startswith("## "):
            section = line[3:].strip()
            continue
        if section != "grid":
            continue
        m = re.match(r"^(\S.*?) dispatches (\d+)", line)          # (a template kernel's name has its arguments: "k_fpc32_sweep<false, true>")
        if m:
            kernel = m.group(1)
            disp[kernel] = int(m.group(2))
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+) per dispatch", line)
        if m and kernel and mine(kernel):
            (fetch if m.group(1) == "FETCH_SIZE" else write)[kernel] = float(m.group(2)) * disp[kernel]
    launches = 0
    for pre in per_launch_of:
        launches = launches or sum(v for k, v in disp.items() if k.startswith(pre))
    if not fetch or not write or not launches:
        return None
    return int((2.0 * sum(fetch.values()) + sum(write.values())) * 1024.0 / launches)


def pmc_traffic(mesh, live=True):
    """HBM bytes per launch sequence of the float-vertex encoder.  Measured in THIS run when rocprofv3 is there: two child
    processes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, --kernel-trace only beside them) over
    tools/perf_fpc32.py, the same kernels on the same grid vertices; else the newest committed summary under profiles/.
    Returns (bytes, source)."""
    import glob
    import shutil
    import tempfile
    if mesh != "grid":
        return None, None
    failed = ""
    if live and shutil.which("rocprofv3"):
        tmp = tempfile.mkdtemp(prefix="trico_pmc_", dir="/tmp")
        try:
            text = ""
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, c)
                env = dict(os.environ, TMPDIR="/tmp")
                subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "perf_fpc32.py"), "grid"], cwd="/tmp", env=env, timeout=240,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
                text += subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d], capture_output=True, text=True,
                                       check=True).stdout
            f = os.path.join(tmp, "summary.txt")
            open(f, "w").write(text)
            t = _pmc_file_traffic(f)
            if t:
                return t, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/perf_fpc32.py grid"
        except Exception as e:       # no profiler rights, timeout ...: say so, loudly, in the line too, and fall back
            failed = "LIVE PMC PASS FAILED (%s): " % (repr(e)[:160])
            sys.stderr.write("bench.py: %susing the committed summary for roofline.traffic\n" % failed)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fpc32_encode_hbm_traffic_pmc.txt")))
    if not files:
        return None, (failed + "no committed summary either") if failed else None
    return _pmc_file_traffic(files[-1]), failed + os.path.relpath(files[-1], ROOT) + " (committed summary, NOT this run)"


def encoder_variants(W, H):
    """The float-vertex encoder alone (tools/perf_fpc32.py: hipEvent span of its launch sequence over the 50 M vertices, 5 timed
    encodes) as child processes, on both meshes: the library's own choice (the one-sweep coder, k_fpc32_sweep.hip) against the other
    coders it contains - two sweeps with the exchange (round 3's encoder, TRICO_FPC32_SWEEPS=2) and two sweeps with ballots (what a
    flagged stream is coded with again and what the decoders' self-check uses, TRICO_FPC32_XCHG=0).  Same bytes in every case
    (tests/test_gpu_onesweep.py); `frac` is algorithmic bytes / span / 8 TB/s like the roofline block."""
    import re
    rows = []
    for mesh in ("grid", "walk"):
        for coder, add in (("default", {}), ("two sweeps, exchange", {"TRICO_FPC32_SWEEPS": "2"}), ("two sweeps, ballots", {"TRICO_FPC32_XCHG": "0"})):
            env = dict(os.environ)
            if add:
                hooks = os.path.join(ROOT, "tests", "_build", "libtrico_testhooks.so")      # the coder switches exist in the test-hooks build only
                if not os.path.exists(hooks):
                    rows.append({"mesh": mesh, "coder": coder, "skipped": "tests/_build/libtrico_testhooks.so not built"})
                    continue
                env["TRICO_AMD_LIB"] = hooks
            env.update(add)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "perf_fpc32.py"), mesh, str(W), str(H)], env=env, timeout=240,
                                   capture_output=True, text=True)
                m = re.search(r"kernel span avg ([0-9.]+) ms; raw ([0-9.]+) MB comp ([0-9.]+) MB", r.stdout)
                if r.returncode != 0 or not m:
                    rows.append({"mesh": mesh, "coder": coder, "error": (r.stdout + r.stderr)[-200:]})
                    continue
                ms, raw, comp = float(m.group(1)), float(m.group(2)), float(m.group(3))
                gb = (raw + comp) / 1e3 / (ms * 1e-3)
                rows.append({"mesh": mesh, "coder": coder, "avg_launch_ms": ms, "achieved": round(gb, 1), "unit": "GB/s",
                             "frac": round(gb / HBM_PEAK_GBPS, 5)})
            except Exception as e:      # noqa: BLE001
                rows.append({"mesh": mesh, "coder": coder, "error": repr(e)[:200]})
    return {"what": "float-vertex encoder alone on both meshes: the library's choice (one sweep) against the two-sweep coders it also "
                    "contains (environment switches, measurements only); traffic of the default on both meshes: profiles/r05_fpc32_encode_hbm_traffic_pmc.txt",
            "rows": rows}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mesh", default="grid", choices=["grid", "walk"])
    ap.add_argument("--W", type=int, default=10000)
    ap.add_argument("--H", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="full", help="WxH of the CPU baseline's mesh; 'full' = the GPU's own mesh")
    ap.add_argument("--concurrent", default="8,16,32,64", help="archives decoded as one batch in the decode_concurrent block ('' : skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the walk mesh, PCIe-inclusive and concurrent-decode blocks")
    ap.add_argument("--quick", action="store_true", help="headline, roofline and the 1-thread CPU baseline only")
    ap.add_argument("--mode", default="roundtrip", choices=["roundtrip", "decode-mixed"],
                    help="'roundtrip' = the headline step (encode + gather + decode of one mesh per GPU); 'decode-mixed' = BASELINE configs[4]: rank r "
                         "decodes an archive of kind (grid, walk, multi)[r mod 3] that it built itself, no collective, GB/s of decoded bytes")
    ap.add_argument("--shard", default="meshes", choices=["meshes", "streams"],
                    help="N > 1: 'meshes' = one mesh per GPU (BASELINE configs[3], weak scaling); 'streams' = ONE mesh, its component "
                         "streams and byte planes spread over the GPUs and assembled into one archive on rank 0 (strong scaling)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "c"],
                    help="transport of the gather: torch.distributed (nccl = RCCL) or the RCCL entry of the C-ABI (trico_hip_comm_*)")
    # the rank body on a host without GPUs (tests/test_bench_ranks_host.py): gloo over CPU tensors, and a stand-in for
    # trico_amd.api.  Never a measurement: the line says so.
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--api", default="trico_amd.api", help=argparse.SUPPRESS)
    return ap.parse_args()


def _reference_streams_parallel(L, v, t, raw, K=1, reps=3):
    """SURVEY 8(d) mode (ii): the only parallelism the format offers a CPU - one thread per independent stream (three
    coordinate streams, four index byte planes) of K archives, 7 K threads, each running the reference's coder on its stream
    (ctypes releases the GIL).  The de-interleaving into those streams is done before the clock starts; every thread touches
    its output buffers before a common start barrier (pre-faulted); best of `reps` repetitions."""
    import threading
    n, p = v.size // 3, t.size
    comps = [np.ascontiguousarray(v.reshape(-1, 3)[:, c]) for c in range(3)]
    planes = [np.ascontiguousarray(t.view(np.uint8).reshape(-1, 4)[:, k]) for k in range(4)]
    cap = L.LZ4_compressBound(p)
    nthreads = 7 * K
    lz = [[np.empty(cap, np.uint8) for _ in range(4)] for _ in range(K)]
    back_lz = [[np.empty(p, np.uint8) for _ in range(4)] for _ in range(K)]
    lz_len = [[0] * 4 for _ in range(K)]
    fp_out = [[ctypes.c_void_p() for _ in range(3)] for _ in range(K)]
    fp_len = [[ctypes.c_uint32() for _ in range(3)] for _ in range(K)]
    back_fp = [[ctypes.c_void_p() for _ in range(3)] for _ in range(K)]
    back_n = [[ctypes.c_uint32() for _ in range(3)] for _ in range(K)]
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]

    def run(jobs):
        gate = threading.Barrier(len(jobs) + 1)

        def wrap(prep, work):
            def f():
                prep()
                gate.wait()
                work()
            return f
        th = [threading.Thread(target=wrap(pr, wk)) for pr, wk in jobs]
        for x in th:
            x.start()
        gate.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        return time.perf_counter() - t0

    def enc_fp(k, c):
        return (lambda: None), (lambda: L.trico_compress(ctypes.byref(fp_len[k][c]), ctypes.byref(fp_out[k][c]), ctypes.c_void_p(comps[c].ctypes.data),
                                                          ctypes.c_uint32(n), ctypes.c_uint32(4), ctypes.c_uint32(10)))

    def enc_lz(k, q):
        def work():
            lz_len[k][q] = L.LZ4_compress_default(ctypes.c_void_p(planes[q].ctypes.data), ctypes.c_void_p(lz[k][q].ctypes.data), p, cap)
        return (lambda: lz[k][q].fill(0)), work

    def dec_fp(k, c):
        return (lambda: None), (lambda: L.trico_decompress(ctypes.byref(back_n[k][c]), ctypes.byref(back_fp[k][c]), fp_out[k][c]))

    def dec_lz(k, q):
        return (lambda: back_lz[k][q].fill(0)), (lambda: L.LZ4_decompress_safe(ctypes.c_void_p(lz[k][q].ctypes.data),
                                                                                  ctypes.c_void_p(back_lz[k][q].ctypes.data), lz_len[k][q], p))

    enc = dec = None
    for rep in range(reps):
        e = run([enc_fp(k, c) for k in range(K) for c in range(3)] + [enc_lz(k, q) for k in range(K) for q in range(4)])
        d = run([dec_fp(k, c) for k in range(K) for c in range(3)] + [dec_lz(k, q) for k in range(K) for q in range(4)])
        if rep == reps - 1:
            ok = all(back_lz[k][q].tobytes() == planes[q].tobytes() for k in (0, K - 1) for q in range(4))
            for c in range(3):
                ok = ok and ctypes.string_at(back_fp[K - 1][c].value, 4 * n) == comps[c].tobytes()
            assert ok
        for k in range(K):
            for c in range(3):
                libc.free(fp_out[k][c])
                libc.free(back_fp[k][c])
        enc = e if enc is None or e < enc else enc
        dec = d if dec is None or d < dec else dec
    tot = K * raw
    return {"value": round(tot / (enc + dec) / 1e9, 4), "unit": "GB/s", "archives": K, "threads": nthreads, "cores": min(nthreads, os.cpu_count() or 1),
            "encode_GBps": round(tot / enc / 1e9, 4), "decode_GBps": round(tot / dec / 1e9, 4), "best_of": reps,
            "what": "%d archive(s) x 7 independent streams (x, y, z, b1..b4), one thread each calling the reference's coder; buffers "
                    "pre-faulted, threads released together" % K}


def mesh_units(arrays):
    """The independent units the format cuts streams into: one per component of a floating-point stream, one per byte
    plane of an integer stream.  arrays: list of (kind, numpy array) with kind 'vec3' / 'vec2' (interleaved float32 / float64) or
    'ints' (uint32 / uint64)."""
    units = []
    for kind, a in arrays:
        if kind == "ints":
            w = a.dtype.itemsize
            b = a.view(np.uint8).reshape(-1, w)
            units += [("plane", np.ascontiguousarray(b[:, k])) for k in range(w)]
        else:
            ar = 3 if kind == "vec3" else 2
            m = a.reshape(-1, ar)
            units += [("fp%d" % (8 * a.dtype.itemsize), np.ascontiguousarray(m[:, c])) for c in range(ar)]
    return units


def reference_units(L, units, parallel=True, reps=2, check=True):
    """The reference's coders (trico_compress* / trico_decompress*, fpsc.c:86-417, 576-1164; LZ4_compress_default /
    LZ4_decompress_safe of its vendored LZ4) on independent units - one thread per unit, released together, buffers touched before
    (`parallel`), or one unit after the other on one thread.  Returns encode and decode seconds (best of `reps`)."""
    import threading
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    n_u = len(units)
    out = [ctypes.c_void_p() for _ in range(n_u)]
    out_len = [ctypes.c_uint32() for _ in range(n_u)]
    back = [ctypes.c_void_p() for _ in range(n_u)]
    back_n = [ctypes.c_uint32() for _ in range(n_u)]
    lz = [np.empty(L.LZ4_compressBound(a.size), np.uint8) if k == "plane" else None for k, a in units]
    lz_back = [np.empty(a.size, np.uint8) if k == "plane" else None for k, a in units]
    lz_len = [0] * n_u

    def enc(i):
        k, a = units[i]
        if k == "plane":
            def work():
                lz_len[i] = L.LZ4_compress_default(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(lz[i].ctypes.data), a.size, lz[i].size)
            return (lambda: lz[i].fill(0)), work
        if k == "fp32":
            return (lambda: None), (lambda: L.trico_compress(ctypes.byref(out_len[i]), ctypes.byref(out[i]), ctypes.c_void_p(a.ctypes.data),
                                                             ctypes.c_uint32(a.size), ctypes.c_uint32(4), ctypes.c_uint32(10)))
        return (lambda: None), (lambda: L.trico_compress_double_precision(ctypes.byref(out_len[i]), ctypes.byref(out[i]), ctypes.c_void_p(a.ctypes.data),
                                                                          ctypes.c_uint32(a.size), ctypes.c_uint64(20), ctypes.c_uint64(20)))

    def dec(i):
        k, a = units[i]
        if k == "plane":
            return (lambda: lz_back[i].fill(0)), (lambda: L.LZ4_decompress_safe(ctypes.c_void_p(lz[i].ctypes.data), ctypes.c_void_p(lz_back[i].ctypes.data),
                                                                                  lz_len[i], a.size))
        f = L.trico_decompress if k == "fp32" else L.trico_decompress_double_precision
        return (lambda: None), (lambda: f(ctypes.byref(back_n[i]), ctypes.byref(back[i]), out[i]))

    def run(jobs):
        if not parallel:
            for pr, _ in jobs:
                pr()
            t0 = time.perf_counter()
            for _, wk in jobs:
                wk()
            return time.perf_counter() - t0
        gate = threading.Barrier(len(jobs) + 1)

        def wrap(pr, wk):
            def f():
                pr()
                gate.wait()
                wk()
            return f
        th = [threading.Thread(target=wrap(pr, wk)) for pr, wk in jobs]
        for x in th:
            x.start()
        gate.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        return time.perf_counter() - t0

    best_e = best_d = None
    for rep in range(reps):
        e = run([enc(i) for i in range(n_u)])
        d = run([dec(i) for i in range(n_u)])
        if check and rep == reps - 1:
            for i, (k, a) in enumerate(units):
                got = lz_back[i].tobytes() if k == "plane" else ctypes.string_at(back[i].value, a.nbytes)
                assert got == a.tobytes(), "reference round trip of unit %d" % i
        for i, (k, _) in enumerate(units):
            if k != "plane":
                libc.free(out[i])
                libc.free(back[i])
        best_e = e if best_e is None or e < best_e else best_e
        best_d = d if best_d is None or d < best_d else best_d
    return best_e, best_d


def cpu_twin(L, arrays, sample, one_thread=True):
    """The reference on the units of `arrays` (mesh_units): one thread per unit, and - `one_thread` - all units one after the other on
    one thread (the reference's own execution model: its API codes the components of a stream in turn, trico.c:229-260)."""
    units = mesh_units(arrays)
    raw = sum(a.nbytes for _, a in arrays)
    pe, pd = reference_units(L, units, parallel=True)
    row = {"kind": "reference", "sample": sample, "raw_bytes": raw, "nproc": os.cpu_count(),
           "streams_parallel": {"threads": len(units), "cores": min(len(units), os.cpu_count() or 1), "encode_GBps": round(raw / pe / 1e9, 4),
                                "decode_GBps": round(raw / pd / 1e9, 4), "value": round(raw / (pe + pd) / 1e9, 4)}}
    if one_thread:
        se, sd = reference_units(L, units, parallel=False, reps=1)
        row.update({"cores": 1, "encode_GBps": round(raw / se / 1e9, 4), "decode_GBps": round(raw / sd / 1e9, 4),
                    "value": round(raw / (se + sd) / 1e9, 4), "unit": "GB/s"})
    return row


def cpu_baseline(mesh, W, H, v, t, all_cores_K=()):
    """Times the CPU path on the host cores of this box: encode + decode through the reference's API, 1 thread (the
    reference's own execution model), one thread per independent stream (the only parallelism the format offers), and - the
    shape decode_concurrent gives the GPU - K archives at once on min(nproc, 7 K) cores (`all_cores`)."""
    from oracle import oracle as O
    nv, nt = W * H, 2 * W * H
    raw = v.nbytes + t.nbytes
    if O.have_ref():
        L = O.ref()
        kind = "reference"
        t0 = time.perf_counter()
        a = L.trico_open_archive_for_writing(1 << 20)
        L.trico_write_vertices(a, v.ctypes.data, nv)
        L.trico_write_triangles(a, t.ctypes.data, nt)
        t1 = time.perf_counter()
        size = L.trico_get_size(a)
        blob = ctypes.string_at(L.trico_get_buffer_pointer(a), size)
        L.trico_close_archive(a)
        buf = np.frombuffer(blob, np.uint8)
        v2, t2 = np.empty_like(v), np.empty_like(t)
        t2s = time.perf_counter()
        r = L.trico_open_archive_for_reading(buf.ctypes.data, size)
        pv, pt = ctypes.c_void_p(v2.ctypes.data), ctypes.c_void_p(t2.ctypes.data)
        L.trico_read_vertices(r, ctypes.byref(pv))
        L.trico_read_triangles(r, ctypes.byref(pt))
        t3 = time.perf_counter()
        L.trico_close_archive(r)
        assert v2.tobytes() == v.tobytes() and t2.tobytes() == t.tobytes()
        enc, dec = t1 - t0, t3 - t2s
        par = _reference_streams_parallel(L, v, t, raw)
        allc = []
        for K in all_cores_K:
            need = K * (4 * (t.size + t.size // 200) + 4 * t.size + 2 * v.nbytes) * 1.3      # LZ4 out + planes back + fp out/back
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except Exception:
                avail = 0
            if (os.cpu_count() or 1) < 7 * K or avail < need:
                allc.append({"archives": K, "skipped": "needs %d cores and %.0f GB of host memory (box: %d cores, %.0f GB available)"
                                                       % (7 * K, need / 1e9, os.cpu_count() or 1, avail / 1e9)})
                continue
            allc.append(_reference_streams_parallel(L, v, t, raw, K=K))
    else:
        kind = "port"
        t0 = time.perf_counter()
        a = O.OracleArchive()
        a.write("vertices", v, nv)
        a.write("triangles", t, nt)
        t1 = time.perf_counter()
        blob = a.tobytes()
        a.close()
        # decode leg of the port: per-payload decoders
        import struct
        pos = 8 + 5
        t2s = time.perf_counter()
        for _ in range(3):
            nb = struct.unpack_from("<I", blob, pos)[0]
            O.fpc_decode(blob[pos + 4: pos + 4 + nb], np.float32)
            pos += 4 + nb
        pos += 5
        for _ in range(4):
            nb = struct.unpack_from("<I", blob, pos)[0]
            O.lz4_decompress(blob[pos + 4: pos + 4 + nb], 3 * nt)
            pos += 4 + nb
        t3 = time.perf_counter()
        enc, dec = t1 - t0, t3 - t2s
        par = None
        allc = []
    return {"streams_parallel": par, "all_cores": allc, "nproc": os.cpu_count(),
            "value": round(raw / (enc + dec) / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
            "sample": "%s(%d,%d): %d float vertices + %d u32 triangles, %.0f MB raw; encode %.2f s, decode %.2f s"
                      % (mesh, W, H, nv, nt, raw / 1e6, enc, dec),
            "encode_GBps": round(raw / enc / 1e9, 4), "decode_GBps": round(raw / dec / 1e9, 4)}


def extras(args, api, meshgen, dev, d_v, d_t, nv, nt, raw_bytes):
    """Blocks reported beside the headline (rank 0, N = 1, outside the timed region): the same step on the other synthetic mesh,
    the step through HOST pointers (PCIe staging included; never `value`), and BASELINE config 5's shape on one GPU: several
    archives decoded at once from host threads, GB/s of decoded bytes."""
    import threading
    out = {}
    W, H = args.W, args.H

    def one_step(dv, dt):
        t0 = time.perf_counter()
        a = api.Archive.open_for_writing(raw_bytes // 2, device=True)
        assert a.write("vertices", dv, nv) == 1 and a.write("triangles", dt, nt) == 1, api.last_error()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o_v, o_t = torch.empty_like(dv), torch.empty_like(dt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
        assert r.read("vertices", o_v) == 1 and r.read("triangles", o_t) == 1, api.last_error()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ok = bool(torch.equal(o_v.view(torch.int32), dv.view(torch.int32)) and torch.equal(o_t, dt))
        size = a.get_size()
        r.close()
        a.close()
        return t1 - t0, t3 - t2, size, ok

    # ---- the other mesh -------------------------------------------------------------------------------------------
    other = "walk" if args.mesh == "grid" else "grid"
    ov, ot = (meshgen.walk if other == "walk" else meshgen.grid)(W, H)
    d_ov, d_ot = torch.from_numpy(ov).to(dev), torch.from_numpy(ot.view(np.int32)).to(dev)
    one_step(d_ov, d_ot)
    e, d, size, ok = one_step(d_ov, d_ot)
    assert ok
    out["other_mesh"] = {"workload": "%s(%d,%d), same sizes" % (other, W, H), "archive_bytes": size, "ratio": round(raw_bytes / size, 2),
                         "value": round(raw_bytes / (e + d) / 1e9, 4), "encode_GBps": round(raw_bytes / e / 1e9, 4),
                         "decode_GBps": round(raw_bytes / d / 1e9, 4)}
    del d_ov, d_ot
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        if O.have_ref():
            tw = cpu_baseline(other, W, H, ov, ot)                # the reference's API on one thread + one thread per stream, the whole mesh
            out["other_mesh"]["cpu_twin"] = {k: tw[k] for k in ("kind", "sample", "cores", "value", "encode_GBps", "decode_GBps", "streams_parallel", "nproc")}
    # ---- host pointers through the plain C API (H2D of the raw arrays, D2H of archive and results included) ---------------
    hv, ht = d_v.cpu().numpy(), d_t.cpu().numpy().view(np.uint32)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        a = api.Archive.open_for_writing(raw_bytes // 4)
        assert a.write("vertices", hv, nv) == 1 and a.write("triangles", ht, nt) == 1, api.last_error()
        t1 = time.perf_counter()
        blob = a.tobytes()
        a.close()
        v2, t2 = np.empty_like(hv), np.empty_like(ht)
        t2s = time.perf_counter()
        r = api.Archive.open_for_reading(blob)
        assert r.read("vertices", v2) == 1 and r.read("triangles", t2) == 1, api.last_error()
        r.close()
        t3 = time.perf_counter()
        if best is None or (t1 - t0) + (t3 - t2s) < best[0] + best[1]:
            best = (t1 - t0, t3 - t2s)
    assert v2.tobytes() == hv.tobytes() and t2.tobytes() == ht.tobytes()
    out["pcie_inclusive"] = {"what": "same mesh through the plain C API with host pointers (pageable numpy arrays)",
                             "value": round(raw_bytes / (best[0] + best[1]) / 1e9, 4), "encode_GBps": round(raw_bytes / best[0] / 1e9, 4),
                             "decode_GBps": round(raw_bytes / best[1] / 1e9, 4)}
    del blob, v2, t2
    # ---- several archives decoded as ONE batch (BASELINE configs[4] on one GPU) -------------------------------------------
    Ks = [int(k) for k in args.concurrent.split(",") if k.strip()]
    if Ks:
        from bench_batch_decode import batch_rows
        rows = batch_rows(Ks, d_v, d_t, nv, nt, raw_bytes, passes=2)
        # the same batch with the hardware queue count of the process set to 1, 4 and 32: child processes (the variable is read
        # when HIP initialises); round 2's one-launch-per-stream decoder went from 7 GB/s to wrong results between the last two
        variants = []
        kq = 32 if 32 in Ks else Ks[-1]
        for q in ("1", "4", "32"):
            env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_batch_decode.py"), "--mesh", args.mesh, "--W", str(W),
                                    "--H", str(H), "--K", str(kq), "--passes", "1"], env=env, capture_output=True, text=True, timeout=300)
                line = [x for x in r.stdout.splitlines() if x.startswith("{")]
                variants.append({"GPU_MAX_HW_QUEUES": q, **json.loads(line[-1])["rows"][0]} if line else {"GPU_MAX_HW_QUEUES": q, "error": r.stderr[-300:]})
            except Exception as e:
                variants.append({"GPU_MAX_HW_QUEUES": q, "error": str(e)})
        # the reference's usage: one host thread per archive, plain trico_read_* calls; the engine combines calls that arrive
        # together into one batch
        import threading
        kt = 8
        a = api.Archive.open_for_writing(raw_bytes // 2, device=True)
        assert a.write("vertices", d_v, nv) == 1 and a.write("triangles", d_t, nt) == 1, api.last_error()
        touts = [(torch.empty_like(d_v), torch.empty_like(d_t)) for _ in range(kt)]
        terr = []

        def tdecode(k):
            r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
            if not (r.read("vertices", touts[k][0]) == 1 and r.read("triangles", touts[k][1]) == 1):
                terr.append(k)
            r.close()

        tbest = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=tdecode, args=(k,)) for k in range(kt)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tbest = dt if tbest is None or dt < tbest else tbest
        assert not terr, terr
        for k in range(kt):
            assert torch.equal(touts[k][0].view(torch.int32), d_v.view(torch.int32)) and torch.equal(touts[k][1], d_t)
        a.close()
        del touts
        threads_row = {"archives": kt, "host_threads": kt, "seconds": round(tbest, 3), "decode_GBps": round(kt * raw_bytes / tbest / 1e9, 3),
                       "what": "one host thread per archive calling trico_read_vertices / trico_read_triangles (round 2's shape); calls that arrive "
                               "together are combined into one batch by the engine"}
        out["decode_concurrent"] = {"threads": threads_row, "what": "K readers of the %s(%d,%d) archive decoded as ONE batch (trico_hip_read_archives: every float chain in one "
                                            "kernel launch, the index streams beside them) on one GPU; GB/s of decoded bytes, outputs compared bit "
                                            "for bit, `repeats` = chain decodes the self-check sent back" % (args.mesh, W, H),
                                    "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "results": rows, "queue_count_variants": variants,
                                    "decode_GBps": max(r["decode_GBps"] for r in rows)}
    return out


def decode_model(api, d_v, d_t, nv, nt, raw_bytes):
    """What bounds the chain decoders: every component chain of the mesh decoded alone (one job of arity 1 per chain through
    trico_hip_decode_jobs, hipEvent span around kernel + self-check), nanoseconds and cycles per value."""
    import struct
    L = api.lib()
    a = api.Archive.open_for_writing(raw_bytes // 2, device=True)
    assert a.write("vertices", d_v, nv) == 1, api.last_error()
    head = bytearray(8 + 5 + 3 * 4 + 64)
    base = a.get_buffer_pointer()
    out = torch.empty(nv, dtype=torch.float32, device=d_v.device)
    ns, pos = {}, 8 + 5
    spans = ctypes.c_uint64(0)
    L.trico_hip_profile_enable(1)
    for name in ("x", "y", "z"):
        sz = (ctypes.c_uint32 * 1)()
        assert L.trico_hip_copy(ctypes.addressof(sz), base + pos, 4) == 1
        jobs = api.make_jobs([{"is_int": 0, "arity": 1, "width": 4, "n": nv, "payloads": [(base + pos + 4, sz[0])], "dst": out}])
        best = None
        for _ in range(2):
            L.trico_hip_profile_reset()
            assert L.trico_hip_decode_jobs(jobs, 1) == 1, api.last_error()
            ms = L.trico_hip_profile_ms(api.KERNEL_IDS["fpc32_decode"], ctypes.byref(spans))
            best = ms if best is None or ms < best else best
        assert torch.equal(out.view(torch.int32), d_v.view(torch.int32)[{"x": 0, "y": 1, "z": 2}[name]::3])
        ns[name] = round(best * 1e6 / nv, 2)
        pos += 4 + sz[0]
    L.trico_hip_profile_enable(0)
    a.close()
    del head
    clock_ghz = 2.4
    return {"ns_per_value": ns, "cycles_per_value": {k: round(v * clock_ghz, 1) for k, v in ns.items()}, "clock_GHz_assumed": clock_ghz,
            "chains": "one per component stream (fpsc.c:308-326: value i needs value i-1 through both tables); the step time is the slowest "
                      "chain's n x ns_per_value, whatever the GPU has left",
            "bound": "issue + scalar-cache latency of ONE wave: straight-line bodies per pattern of kinds - 11 scalar instructions at 4 cycles "
                     "for an FCM-coded value before an FCM-coded one, 15 + the table load's ~37-cycle latency on the dependent path for DFCM-coded "
                     "ones, ~9 cycles per value to dispatch a quad (DESIGN.md 4.5); smooth streams skip batches of exact hits",
            "algorithmic_GBps_per_chain": {k: round(4.0 / v, 3) for k, v in ns.items()}}


def config5_mixed(api, meshgen, dev, W, H, multi_devs, grid_dev, with_cpu=True):
    """BASELINE configs[4] on ONE GPU: eight archives of mixed content - 3 x grid (float vertices, u32 triangles), 3 x walk (the
    same types, noisy), 2 x multi (double vertices + normals, float uv, u64 triangles) - decoded as ONE batch, GB/s of decoded bytes."""
    wv, wt = meshgen.walk(W, H)
    walk_dev = [("vertices", torch.from_numpy(wv).to(dev), W * H), ("triangles", torch.from_numpy(wt.view(np.int32)).to(dev), 2 * W * H)]
    del wv, wt
    kinds = [("grid", grid_dev), ("walk", walk_dev), ("multi", multi_devs)]
    arch = {}
    for name, devs in kinds:
        raw = sum(d.numel() * d.element_size() for _, d, _ in devs)
        a = api.Archive.open_for_writing(raw // 2, device=True)
        for sname, d, cnt in devs:
            assert a.write(sname, d, cnt) == 1, api.last_error()
        arch[name] = (a, devs, raw)
    order = ["grid", "walk", "multi", "grid", "walk", "multi", "grid", "walk"]
    outs = [[torch.empty_like(d) for _, d, _ in arch[k][1]] for k in order]
    best = None
    for _ in range(2):
        readers = [api.Archive.open_for_reading(arch[k][0].get_buffer_pointer(), arch[k][0].get_size()) for k in order]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = api.read_archives(readers, outs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert ok == 1, api.last_error()
        for r in readers:
            r.close()
        best = dt if best is None or dt < best else best
    for k, o in zip(order, outs):
        for (_, d, _), got in zip(arch[k][1], o):
            assert torch.equal(got.view(torch.uint8), d.view(torch.uint8))
    total = sum(arch[k][2] for k in order)
    for a, _, _ in arch.values():
        a.close()
    twin = None
    if with_cpu:
        from oracle import oracle as O
        if O.have_ref():
            # the reference on the same batch, one thread per independent stream of every archive (3 x 7 + 3 x 7 + 2 x 16 = 74), on meshes of
            # a quarter of the vertices: decode time only (the payloads are made by the same threads first)
            sw, sh = max(16, W // 2), max(16, H // 2)
            gv, gt = meshgen.grid(sw, sh)
            wv2, wt2 = meshgen.walk(sw, sh)
            mv, mn, muv, mt = meshgen.multi(sw, sh)
            arrays = 3 * [("vec3", gv), ("ints", gt)] + 3 * [("vec3", wv2), ("ints", wt2)] + 2 * [("vec3", mv), ("vec3", mn), ("vec2", muv), ("ints", mt)]
            units = mesh_units(arrays)
            raw = sum(a.nbytes for _, a in arrays)
            _, pd = reference_units(O.ref(), units, parallel=True, reps=2, check=False)
            twin = {"kind": "reference", "sample": "the same 8 archives with meshes of %d x %d (a quarter of the vertices), %d decoded bytes" % (sw, sh, raw),
                    "threads": len(units), "cores": min(len(units), os.cpu_count() or 1), "nproc": os.cpu_count(),
                    "decode_GBps": round(raw / pd / 1e9, 3), "unit": "GB/s"}
    return {"workload": "8 archives decoded as one batch: 3 x grid + 3 x walk (float vertices, u32 triangles) + 2 x multi (double vertices, double "
                        "normals, float uv, u64 triangles), %d x %d each (BASELINE configs[4] on one GPU)" % (W, H),
            "decoded_bytes": total, "seconds": round(best, 3), "decode_GBps": round(total / best / 1e9, 3),
            "chains": {"float": 3 * 6 + 2 * 2, "double": 2 * 6}, "bound": "the noisy double chains of the multi archives (decode_s of config3)",
            "cpu_twin": twin}


def _component_sizes(blob, ncomp_per_stream):
    """payload sizes of the streams of an archive, in order (framing of trico.c:215-262: u8 type, u32 count, then u32 size + payload per component)"""
    import struct
    pos, out = 8, []
    for nc in ncomp_per_stream:
        pos += 5
        sizes = []
        for _ in range(nc):
            nb = struct.unpack_from("<I", blob, pos)[0]
            sizes.append(nb)
            pos += 4 + nb
        out.append(sizes)
    return out


def config3_block(api, meshgen, dev, W, H, grid_dev=None, with_cpu=True):
    """BASELINE configs[2]: double vertices + double normals + float uv (+ u64 triangles: the archive is then the reference's
    golden multi_WxH), device-resident encode and decode; sha256 against tests/golden/hashes.json."""
    v, nrm, uv, t = meshgen.multi(W, H)
    n = W * H
    streams = [("vertices_double", v, n), ("vertex_normals_double", nrm, n), ("uv_per_vertex", uv, n), ("triangles_long", t, 2 * n)]
    devs = [(name, torch.from_numpy(a.view(np.uint8)).to(dev), cnt) for name, a, cnt in streams]
    raw = sum(a.nbytes for _, a, _ in streams)
    fp_raw = v.nbytes + nrm.nbytes + uv.nbytes
    del v, nrm, uv, t
    res = None
    L = api.lib()
    spans = ctypes.c_uint64(0)
    enc64_ms = None
    for rep in range(2):
        a = api.Archive.open_for_writing(raw // 3, device=True)
        torch.cuda.synchronize()
        L.trico_hip_profile_enable(1)
        L.trico_hip_profile_reset()
        t0 = time.perf_counter()
        for name, d, cnt in devs:
            assert a.write(name, d, cnt) == 1, api.last_error()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ms = L.trico_hip_profile_ms(api.KERNEL_IDS["fpc64_encode"], ctypes.byref(spans))
        L.trico_hip_profile_enable(0)
        if spans.value:
            enc64_ms = ms                                    # device time of the launch sequences of the two vec3 double streams together
        outs = [torch.empty_like(d) for _, d, _ in devs]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
        for (name, d, cnt), o in zip(devs, outs):
            assert r.read(name, o) == 1, api.last_error()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ok = all(bool(torch.equal(o, d)) for (_, d, _), o in zip(devs, outs))
        size = a.get_size()
        if rep == 0:
            blob = a.tobytes()
            sha = hashlib.sha256(blob).hexdigest()
            comp_sizes = _component_sizes(blob, [3, 3, 2, 8])
            del blob
        else:
            sha = res["sha256"]
        r.close()
        a.close()
        del outs
        res = {"sha256": sha, "archive_bytes": size, "encode_s": round(t1 - t0, 4), "decode_s": round(t3 - t2, 4), "roundtrip_ok": ok}
    mixed = None
    if grid_dev is not None:
        mixed = config5_mixed(api, meshgen, dev, W, H, devs, grid_dev, with_cpu=with_cpu)
    # the double encoder against the HBM roof: algorithmic bytes (raw doubles in + payload bytes out) / device time
    roof = None
    dbl_raw = 2 * 3 * 8 * n
    dbl_pay = sum(comp_sizes[0]) + sum(comp_sizes[1])
    if enc64_ms:
        gb = (dbl_raw + dbl_pay) / (enc64_ms * 1e-3) / 1e9
        # traffic: not collected live; the committed PMC summary of the same two streams at full size (profiles/r05_fpc64_encode_hbm_traffic_pmc.txt:
        # FETCH_SIZE x 2 + WRITE_SIZE over the kernels of both streams), scaled by the value count when the mesh is another
        full = 50_000_000
        import glob
        t64, src64 = None, "no committed profiles/*_fpc64_encode_hbm_traffic_pmc.txt"
        files64 = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fpc64_encode_hbm_traffic_pmc.txt")))
        files64 = [f for f in files64 if "_before_" not in f]
        if files64:
            per_stream = _pmc_file_traffic(files64[-1], family=None, per_launch_of=("k64_sizes",))
            if per_stream:
                t64 = int(2 * per_stream * n / full)
                src64 = os.path.relpath(files64[-1], ROOT) + " (separate --pmc passes over tools/perf_fpc64.py at full size, NOT this run; scaled by the value count)"
        ratio64 = ("%.0f x" % (t64 / (dbl_raw + dbl_pay))) if t64 else "many times"
        roof = {"kernel": "double encoder (k_fpc64_sort.hip: runs of equal hashes, 1024 owners per component and table, operations partitioned "
                          "into their lists, LDS walk, results home by tile, codes, scan, pack), the two vec3 double streams of the mesh",
                "bound": "hbm", "achieved": round(gb, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gb / HBM_PEAK_GBPS, 5), "traffic": t64,
                "traffic_source": src64,
                "algorithmic_bytes": dbl_raw + dbl_pay, "launch_ms": round(enc64_ms, 3),
                "what_bounds_it": "memory traffic " + ratio64 + " the algorithmic bytes at ~3 TB/s: a table operation is 16 bytes written into one of 3072 lists per "
                                  "wave (no line is written whole: the counters show twice the bytes), read by the walk, written as a result, read "
                                  "again by tile; predictions are written once and read by the size and the pack kernel"}
    twin = None
    if with_cpu:
        from oracle import oracle as O
        if O.have_ref():
            sw, sh = max(16, W // 4), max(16, (2 * H) // 5)                    # a tenth of the vertices: ~20 s of host time
            mv, mn, muv, mt = meshgen.multi(sw, sh)
            twin = cpu_twin(O.ref(), [("vec3", mv), ("vec3", mn), ("vec2", muv), ("ints", mt)],
                            "multi(%d,%d): %d double vertices + double normals + float uv + u64 triangles (a tenth of the block's mesh); the units of a "
                            "stream one after the other on one thread, and one thread per unit (6 double + 2 float components, 8 byte planes)" % (sw, sh, sw * sh))
    g = None
    hp = os.path.join(ROOT, "tests", "golden", "hashes.json")
    if os.path.exists(hp):
        g = json.load(open(hp)).get("multi_%dx%d" % (W, H))
    assert res["roundtrip_ok"]
    parity = "unchecked (no golden for this size)"
    if g is not None:
        if g["sha256"] != res["sha256"]:
            raise SystemExit("bench.py: config-3 archive sha256 differs from the reference's golden")
        parity = "sha256 == reference golden"
    return {"workload": "multi(%d,%d): %d double vertices + double normals + float uv (BASELINE configs[2]) + u64 triangles (the reference's "
                        "golden archive of this mesh), device-resident" % (W, H, n),
            "raw_bytes": raw, "floating_point_raw_bytes": fp_raw, "archive_bytes": res["archive_bytes"], "parity": parity,
            "encode_GBps": round(raw / res["encode_s"] / 1e9, 3), "decode_GBps": round(raw / res["decode_s"] / 1e9, 3),
            "encode_s": res["encode_s"], "decode_s": res["decode_s"],
            "value": round(raw / (res["encode_s"] + res["decode_s"]) / 1e9, 4), "roofline": roof, "cpu_twin": twin, "config5_mixed": mixed}


def rank_report(dist, rank, local_rank, world, dev, host_only, facts):
    """What a reader of an N > 1 line needs to see that N ranks really ran on N devices: per rank its device ordinal, PCI bus id, name
    and the facts it hands in (archive verdict, bytes); and what the collective backend itself says about the job (backend name, its
    world size, the RCCL version).  Gathered with all_gather_object outside the timed region; returned on every rank."""
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
    if not host_only:
        pr = torch.cuda.get_device_properties(dev)
        me.update(device_ordinal=dev.index, device=pr.name, cus=pr.multi_processor_count,
                  pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                  uuid=str(getattr(pr, "uuid", "")))
    me.update(facts)
    rows = [me]
    backend = {"backend": None, "world_size_as_reported": 1}
    if dist is not None:
        rows = [None] * world
        dist.all_gather_object(rows, me)
        backend = {"backend": dist.get_backend(), "world_size_as_reported": dist.get_world_size()}
        if not host_only:
            try:
                backend["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:
                backend["rccl_version"] = "unknown (%s)" % repr(e)[:80]
    distinct = len(set((r.get("pci"), r.get("device_ordinal")) for r in rows)) if not host_only else None
    return {"collective": backend, "distinct_devices": distinct, "per_rank": rows}


def _gathered_bytes(g):
    """bytes the exchange delivered to this rank: (tensor, sizes) from gather_archives on the root, a count from the C-ABI gather, None elsewhere"""
    if g is None:
        return 0
    if isinstance(g, tuple):
        return int(sum(g[1]))
    try:
        return int(g)
    except Exception:
        return int(getattr(g, "numel", lambda: 0)())


def _headline_last(out):
    """The driver keeps the TAIL of stdout: the keys of the contract go to the end of the line, the bulky blocks in front of them."""
    head = ["ranks", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "encode_GBps", "decode_GBps", "gather_ms", "roofline", "cpu_baseline"]
    ordered = {k: v for k, v in out.items() if k not in head}
    ordered.update({k: out[k] for k in head if k in out})
    return ordered


def decode_mixed(args, api, meshgen, dev, sync, dist, rank, world, real_stdout, host_only):
    """BASELINE configs[4] as the N-GPU job: every rank decodes its own archive, kinds mixed over the ranks (rank r: grid, walk, multi,
    grid, ...): float and double components, u32 and u64 index planes.  The archive is built by the rank itself before the clock
    starts (and compared with the reference's golden where there is one); a step = open the archive, read every stream into HBM.
    No data-path collective: ranks only meet at the barriers around the timed region."""
    W, H = args.W, args.H
    n = W * H
    kind = ("grid", "walk", "multi")[rank % 3]
    if kind == "multi":
        v, nrm, uv, t = meshgen.multi(W, H)
        streams = [("vertices_double", v, n), ("vertex_normals_double", nrm, n), ("uv_per_vertex", uv, n), ("triangles_long", t, 2 * n)]
    else:
        v, t = (meshgen.grid if kind == "grid" else meshgen.walk)(W, H)
        streams = [("vertices", v, n), ("triangles", t, 2 * n)]
    devs = [(name, torch.from_numpy(a.view(np.uint8)).to(dev), cnt) for name, a, cnt in streams]
    raw = sum(a.nbytes for _, a, _ in streams)
    a = api.Archive.open_for_writing(raw // 3, device=True)
    for name, d, cnt in devs:
        assert a.write(name, d, cnt) == 1, api.last_error()
    sync()
    sha = hashlib.sha256(a.tobytes()).hexdigest()
    parity = "unchecked (no golden for this size)"
    hp = os.path.join(ROOT, "tests", "golden", "hashes.json")
    if os.path.exists(hp):
        g = json.load(open(hp)).get("%s_%dx%d" % (kind, W, H))
        if g is not None:
            if g["sha256"] != sha:
                raise SystemExit("bench.py: %s archive sha256 differs from the reference's golden on rank %d" % (kind, rank))
            parity = "sha256 == reference golden"
    outs = [torch.empty_like(d) for _, d, _ in devs]

    def step():
        r = api.Archive.open_for_reading(a.get_buffer_pointer(), a.get_size())
        for (name, d, cnt), o in zip(devs, outs):
            assert r.read(name, o) == 1, api.last_error()
        sync()
        r.close()

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(max(1, args.warmup)):
        step()
    if not all(bool(torch.equal(o, d)) for (_, d, _), o in zip(devs, outs)):
        raise SystemExit("bench.py: decoded arrays differ from the input on rank %d" % rank)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(raw)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    ranks = rank_report(dist, rank, int(os.environ.get("LOCAL_RANK", "0")), world, dev, host_only,
                        {"kind": kind, "archive_bytes": int(a.get_size()), "archive_sha256": sha[:16], "parity": parity, "decoded_bytes": int(raw),
                         "rank_ms_per_step": round((t1 - t0) / args.steps * 1e3, 3)})
    if rank == 0:
        step_s = el.item() / args.steps
        out = {"metric": "decode GB/s (output bytes), mixed archives", "value": round(tot.item() / step_s / 1e9, 4), "unit": "GB/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u32 / u64 bit patterns",
               "data": "synthetic" if not host_only else "synthetic; HOST REHEARSAL of the rank body (gloo, stand-in coder): not a measurement",
               "config": {"workload": "BASELINE configs[4]: one .trc archive per GPU, kinds (grid, walk, multi)[rank mod 3] of %d x %d - float vertices + u32 "
                                      "triangles / double vertices + double normals + float uv + u64 triangles - built on the GPU, decoded into HBM" % (W, H),
                          "mode": "decode-mixed", "kind_rank0": kind, "decoded_bytes_all_ranks": int(tot.item()), "parity_rank0": parity,
                          "parallelism": "1 process per GPU, %d independent archives, no collective in the timed region" % world},
               "decode_GBps": round(tot.item() / step_s / 1e9, 4), "ranks": ranks}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(_headline_last(out)) + "\n").encode())
    a.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    # RCCL prints a version banner on stdout when it initialises; the driver wants exactly one JSON line there.  Everything
    # this process (and the libraries it loads) writes to fd 1 goes to stderr; the JSON line is written to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    host_only = args.backend == "gloo"
    if not host_only and (local_rank >= torch.cuda.device_count() or world > torch.cuda.device_count()):
        # under a launcher: the same check as _spawn_ranks_if_needed, before the rendezvous
        sys.stderr.write("bench.py: rank %d of %d has no device of its own (%d visible): not started\n" % (rank, world, torch.cuda.device_count()))
        sys.exit(2)
    if world > 1 or args.shard == "streams":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_only:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if host_only:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize

    import importlib
    api = importlib.import_module(args.api)
    from trico_amd import meshgen
    from trico_amd.parallel import gather_archives, wrap_device_bytes, sharded_write, hip_unit_encoder, CComm
    L = api.lib()
    if not L.trico_hip_available():
        raise SystemExit("bench.py: no HIP device: " + api.last_error())
    if args.mode == "decode-mixed":
        if dist is None and world > 1:
            raise SystemExit("bench.py: no process group")
        return decode_mixed(args, api, meshgen, dev, sync, dist, rank, world, real_stdout, host_only)

    W, H = args.W, args.H
    nv, nt = W * H, 2 * W * H
    sharded = args.shard == "streams"
    seed = (meshgen.GRID_SEED if args.mesh == "grid" else meshgen.WALK_SEED) + (0 if sharded else rank)
    gen = meshgen.grid if args.mesh == "grid" else meshgen.walk
    v, t = gen(W, H, seed)
    d_v = torch.from_numpy(v).to(dev)
    d_t = torch.from_numpy(t.view(np.int32)).to(dev)
    d_v2 = torch.empty_like(d_v)
    d_t2 = torch.empty_like(d_t)
    raw_bytes = v.nbytes + t.nbytes

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    state = {}
    comm = None
    if dist is not None and args.exchange == "c":
        def share(b):
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = CComm(api, rank, world, share)
    unit_encoder = (getattr(api, "unit_encoder", None) or hip_unit_encoder)(api) if sharded else None

    def step_sharded(check=False):
        """ONE mesh: every rank encodes its share of the 7 units (x, y, z, b1..b4), one exchange, rank 0 frames the archive and
        decodes it."""
        te0 = time.perf_counter()
        a = sharded_write(dist, api, [("vertices", d_v, nv), ("triangles", d_t, nt)], unit_encoder, root=0, device_archive=True,
                          gather=(lambda t: comm.gather(t, 0)) if comm is not None else None)
        sync()
        te1 = time.perf_counter()
        td1 = te1
        if rank == 0:
            size = a.get_size()
            r = api.Archive.open_for_reading(a.get_buffer_pointer(), size)
            assert r.read("vertices", d_v2) == 1 and r.read("triangles", d_t2) == 1, api.last_error()
            sync()
            td1 = time.perf_counter()
            if check:
                blob = a.tobytes()
                state["archive_bytes"] = len(blob)
                state["sha256"] = hashlib.sha256(blob).hexdigest()
                state["roundtrip_ok"] = bool(torch.equal(d_v2.view(torch.int32), d_v.view(torch.int32)) and torch.equal(d_t2, d_t))
                import struct
                pos, vp = 8 + 5, 0
                for _ in range(3):
                    nb = struct.unpack_from("<I", blob, pos)[0]
                    vp += nb
                    pos += 4 + nb
                state["vertex_payload_bytes"] = vp
            r.close()
            a.close()
        elif check:
            state.update(archive_bytes=0, sha256=None, roundtrip_ok=True, vertex_payload_bytes=0)
        return te1 - te0, 0.0, td1 - te1

    def step(check=False):
        if sharded:
            return step_sharded(check)
        te0 = time.perf_counter()
        a = api.Archive.open_for_writing(raw_bytes // 2, device=True)      # caller-chosen initial size: no regrowth, and room for the vertex stream's worst case (the writer then frames it in place)
        assert a.write("vertices", d_v, nv) == 1, api.last_error()
        assert a.write("triangles", d_t, nt) == 1, api.last_error()
        sync()
        te1 = time.perf_counter()
        size = a.get_size()
        if dist is not None:
            local = wrap_device_bytes(a.get_buffer_pointer(), size, dev)
            gathered = comm.gather(local, 0) if comm is not None else gather_archives(dist, local, dst=0)
            sync()
            state["gathered_bytes"] = gathered
        tg1 = time.perf_counter()
        r = api.Archive.open_for_reading(a.get_buffer_pointer(), size)
        assert r is not None
        assert r.read("vertices", d_v2) == 1, api.last_error()
        assert r.read("triangles", d_t2) == 1, api.last_error()
        sync()
        td1 = time.perf_counter()
        if check:
            blob = a.tobytes()
            state["archive_bytes"] = len(blob)
            state["sha256"] = hashlib.sha256(blob).hexdigest()
            state["roundtrip_ok"] = bool(torch.equal(d_v2.view(torch.int32), d_v.view(torch.int32)) and torch.equal(d_t2, d_t))
            # vertex payload bytes for the roofline's algorithmic byte count
            import struct
            pos, vp = 8 + 5, 0
            for _ in range(3):
                nb = struct.unpack_from("<I", blob, pos)[0]
                vp += nb
                pos += 4 + nb
            state["vertex_payload_bytes"] = vp
        r.close()
        a.close()
        return te1 - te0, tg1 - te1, td1 - tg1

    # correctness gate + warmup (untimed)
    step(check=True)
    for _ in range(max(0, args.warmup - 1)):
        step()
    golden = None
    hp = os.path.join(ROOT, "tests", "golden", "hashes.json")
    if os.path.exists(hp):
        key = "%s_%dx%d" % (args.mesh, W, H) + ("" if (rank == 0 or sharded) else "_seed%08x" % seed)
        golden = json.load(open(hp)).get(key)
    parity = "unchecked (no golden for this size)"
    if golden is not None and state["sha256"] is not None:
        if golden["sha256"] != state["sha256"]:
            raise SystemExit("bench.py: archive sha256 differs from the reference's golden on rank %d" % rank)
        parity = "sha256 == reference golden"
    if not state["roundtrip_ok"]:
        raise SystemExit("bench.py: decoded arrays differ from the input on rank %d" % rank)

    L.trico_hip_profile_enable(1)
    L.trico_hip_profile_reset()
    barrier()
    t0 = time.perf_counter()
    enc = gat = dec = 0.0
    for _ in range(args.steps):
        e, g, d = step()
        enc += e
        gat += g
        dec += d
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0, enc, gat, dec], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = elapsed.tolist()
    ranks = None
    if world > 1 or dist is not None:
        ranks = rank_report(dist, rank, local_rank, world, dev, host_only,
                            {"mesh_seed": "%08x" % seed, "archive_bytes": int(state.get("archive_bytes") or 0), "archive_sha256": (state.get("sha256") or "")[:16],
                             "parity": ("sha256 == reference golden" if (golden is not None and state["sha256"] is not None) else "unchecked (no golden / not the framing rank)"),
                             "roundtrip_ok": bool(state["roundtrip_ok"]), "gathered_bytes_seen": _gathered_bytes(state.get("gathered_bytes")),
                             "rank_ms_per_step": round((t1 - t0) / args.steps * 1e3, 3), "rank_encode_ms": round(enc / args.steps * 1e3, 3),
                             "rank_gather_ms": round(gat / args.steps * 1e3, 3), "rank_decode_ms": round(dec / args.steps * 1e3, 3)})

    spans = ctypes.c_uint64(0)
    kms = {}
    for name, kid in api.KERNEL_IDS.items():
        ms = L.trico_hip_profile_ms(kid, ctypes.byref(spans))
        if spans.value:
            kms[name] = {"avg_ms": round(ms / spans.value, 4), "launches": int(spans.value)}
    L.trico_hip_profile_enable(0)

    if rank == 0:
        total_raw = raw_bytes * (1 if sharded else world)
        step_s = elapsed[0] / args.steps
        fe = kms.get("fpc32_encode", {"avg_ms": float("nan")})
        alg_bytes = v.nbytes + state["vertex_payload_bytes"]
        achieved = alg_bytes / (fe["avg_ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.mesh, live=(world == 1 and not args.quick))
        # every other stage of the step against the same peak: algorithmic bytes / live hipEvent span
        tri_raw = t.nbytes
        tri_pay = state["archive_bytes"] - 8 - 2 * 5 - 7 * 4 - state["vertex_payload_bytes"]
        stage_bytes = {"planes_split": 2 * tri_raw, "lz4_encode": tri_raw + tri_pay, "lz4_decode": tri_raw + tri_pay, "planes_merge": 2 * tri_raw,
                       "fpc32_decode": v.nbytes + state["vertex_payload_bytes"]}
        stage_bound = {"planes_split": "hbm", "planes_merge": "hbm",
                       "lz4_encode": "latency of the greedy parse chain per plane, made parallel by chunk speculation (DESIGN.md 4.3)",
                       "lz4_decode": "hbm traffic of the pointer-jumping rounds, 12 B per output byte and round (DESIGN.md 4.4)",
                       "fpc32_decode": "one dependent chain per component: see decode_model"}
        roofline_all = []
        for name in ("planes_split", "lz4_encode", "fpc32_decode", "lz4_decode", "planes_merge"):
            if name in kms and sharded is False:
                ms = kms[name]["avg_ms"]
                gb = stage_bytes[name] / (ms * 1e-3) / 1e9
                roofline_all.append({"stage": name, "algorithmic_bytes": stage_bytes[name], "avg_ms": ms, "achieved": round(gb, 2), "unit": "GB/s",
                                     "peak": HBM_PEAK_GBPS, "frac": round(gb / HBM_PEAK_GBPS, 5), "bound": stage_bound[name]})
        out = {
            "metric": "encode+decode GB/s (input bytes)",
            "value": round(total_raw / step_s / 1e9, 4),
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic" if not host_only else "synthetic; HOST REHEARSAL of the rank body (gloo, stand-in coder): not a measurement",
            "config": {"workload": "%s(%d,%d): %d float xyz vertices + %d uint32 triangles per GPU (BASELINE configs[1]%s), "
                                   "device-resident raw arrays -> .trc archive in HBM -> decoded arrays in HBM"
                                   % (args.mesh, W, H, nv, nt, "" if world == 1 else "; one mesh per GPU, RCCL gather of archives to rank 0 (configs[3])"),
                       "raw_bytes_per_gpu": raw_bytes, "archive_bytes_rank0": state["archive_bytes"], "parity": parity,
                       "parallelism": ("1 process per GPU, ONE mesh: 7 stream units (x, y, z, b1..b4) round-robin over %d ranks, payload gather, "
                                       "archive assembled and decoded on rank 0" % world) if sharded
                                      else "1 process per GPU, %d independent meshes" % world,
                       "exchange": "RCCL through the C-ABI (trico_hip_comm_gather)" if comm is not None else "torch.distributed (nccl = RCCL)"},
            "encode_GBps": round(total_raw / (elapsed[1] / args.steps) / 1e9, 4),
            "decode_GBps": round(total_raw / (elapsed[3] / args.steps) / 1e9, 4),
            "gather_ms": round(elapsed[2] / args.steps * 1e3, 3),
            "roofline": {"kernel": "float-vertex encoder (every kernel of its launch sequence, AoS vertices -> payload bytes in the archive)",
                         "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fe["avg_ms"],
                         "code_sweep": {0: "two sweeps, ballots", 2: "two sweeps, lane-ordered LDS exchange", 3: "one sweep, lane-ordered LDS exchange"}.get(
                             L.trico_hip_fpc32_code_sweep(), "?")},
            "roofline_all": roofline_all,
            "kernels": kms,
        }
        if ranks is not None:
            out["ranks"] = ranks
        Ks = [int(k) for k in args.concurrent.split(",") if k.strip()]
        if world == 1 and not args.no_extras and not args.quick:
            out["decode_model"] = decode_model(api, d_v, d_t, nv, nt, raw_bytes)
            out.update(extras(args, api, meshgen, dev, d_v, d_t, nv, nt, raw_bytes))
            if args.mesh == "grid":
                del d_v2, d_t2
                torch.cuda.empty_cache()
                out["config3"] = config3_block(api, meshgen, dev, W, H, grid_dev=[("vertices", d_v, nv), ("triangles", d_t, nt)],
                                               with_cpu=not args.no_cpu_baseline)
                out["config5_mixed"] = out["config3"].pop("config5_mixed")
                out["encoder_variants"] = encoder_variants(W, H)
        if world == 1 and not args.no_cpu_baseline:
            allK = tuple(k for k in Ks if k in (8, 16, 32)) if not (args.no_extras or args.quick) else ()
            if args.cpu_sample == "full":
                out["cpu_baseline"] = cpu_baseline(args.mesh, W, H, v, t, allK)
            else:
                cw, ch = (int(x) for x in args.cpu_sample.split("x"))
                cv, ct = gen(cw, ch)
                out["cpu_baseline"] = cpu_baseline(args.mesh, cw, ch, cv, ct, allK)
        # where the GPU wins: K archives decoded as one batch against the reference's coder on 7 K host threads of this box
        if "decode_concurrent" in out and out.get("cpu_baseline", {}).get("all_cores"):
            cpu_rows = {r["archives"]: r for r in out["cpu_baseline"]["all_cores"] if "decode_GBps" in r}
            best = {}
            for r in out["decode_concurrent"]["results"]:
                if r.get("archives") in cpu_rows and r["decode_GBps"] > best.get(r["archives"], 0.0):
                    best[r["archives"]] = r["decode_GBps"]
            out["decode_concurrent"]["vs_cpu_all_cores"] = [
                {"archives": K, "gpu_decode_GBps": g, "cpu_decode_GBps": cpu_rows[K]["decode_GBps"], "cpu_threads": cpu_rows[K]["threads"],
                 "gpu_over_cpu": round(g / cpu_rows[K]["decode_GBps"], 3)} for K, g in sorted(best.items())]
        if "cpu_baseline" in out:
            cb = out["cpu_baseline"]
            out["cpu_baseline_detail"] = {k: cb[k] for k in ("streams_parallel", "all_cores", "nproc", "encode_GBps", "decode_GBps") if k in cb}
            out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(_headline_last(out)) + "\n").encode())
    if unit_encoder is not None:
        unit_encoder.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
