"""In-tree build of the native libraries (gfx950 only).

    python -m trico_amd.build            # build what is stale
    python -m trico_amd.build --force

Outputs (git-ignored, but they travel to the GPU box with the gpurun snapshot):
    trico_amd/lib/libtrico.so            host C container + HIP kernels + C-ABI shim
    trico_amd/lib/libtrico.a             the same objects as a static library (the reference's default flavour, TRICO_SHARED=no)
    trico_amd/lib/libtrico_meshgen.so    synthetic mesh generators (tests / bench only)
Only on request (build(test_hooks=True), what the test suite and __graft_entry__.build() ask for), outside the package:
    tests/_build/libtrico_testhooks.so   the same library + the sabotage switches and the fake RCCL transport of the tests
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
INCLUDE = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CC = os.environ.get("CC", "gcc")
ARCH = "gfx950"
EXTRA = os.environ.get("TRICO_HIPCC_FLAGS", "").split()      # experiments: extra compiler flags for the HIP sources (e.g. -DTRICO_PF=4)

HOST_C = ["host/archive.c", "host/lowlevel.c", "host/lz4_api.c"]
HIP_SRC = sorted(f for f in os.listdir(os.path.join(CSRC, "hip")) if f.endswith(".hip"))
HIP_HDR = sorted(f for f in os.listdir(os.path.join(CSRC, "hip")) if f.endswith((".hpp", ".inc")))      # (.inc: generated chain bodies)

LIBTRICO = os.path.join(LIBDIR, "libtrico.so")
LIBTRICO_A = os.path.join(LIBDIR, "libtrico.a")
HOOKDIR = os.path.join(ROOT, "tests", "_build")
LIBTRICO_HOOKS = os.path.join(HOOKDIR, "libtrico_testhooks.so")   # same library + the sabotage switches (tests only; never in lib/)
HOOKED = ("shim.hip", "dist.hip", "k_fpc32_sweep.hip", "k_fpc32_encode.hip", "k_lz4_chunked.hip")                        # sources that look at TRICO_HIP_TEST_HOOKS
LIBMESHGEN = os.path.join(LIBDIR, "libtrico_meshgen.so")
LIBIO = os.path.join(LIBDIR, "libtrico_io.so")
BINDIR = os.path.join(HERE, "bin")
TOOLS = ["trico_encoder", "trico_decoder"]


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, test_hooks=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, "trico", h) for h in ("trico.h", "trico_hip.h")] + [os.path.join(INCLUDE, "lz4", "lz4.h")]
    headers += [os.path.join(CSRC, "hip", h) for h in HIP_HDR]
    objs = []
    hook_objs = {}
    for rel in HOST_C:
        src = os.path.join(CSRC, rel)
        obj = os.path.join(OBJDIR, os.path.basename(rel) + ".o")
        if force or _stale(obj, [src] + headers):
            _run([CC, "-O2", "-std=c11", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
                  "-I" + INCLUDE, "-c", src, "-o", obj])
        objs.append(obj)
    for f in HIP_SRC:
        src = os.path.join(CSRC, "hip", f)
        obj = os.path.join(OBJDIR, f + ".o")
        if force or _stale(obj, [src] + headers):
            _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                  "-Wall", "-Wno-unused-function", "-I" + INCLUDE, "-I" + os.path.join(CSRC, "hip")] + EXTRA +
                 ["-c", src, "-o", obj])
        objs.append(obj)
        if test_hooks and f in HOOKED:
            hobj = os.path.join(OBJDIR, f + ".hooks.o")
            if force or _stale(hobj, [src] + headers):
                _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DTRICO_HIP_TEST_HOOKS",
                      "-Wall", "-Wno-unused-function", "-I" + INCLUDE, "-I" + os.path.join(CSRC, "hip"),
                      "-c", src, "-o", hobj])
            hook_objs[obj] = hobj
    if force or _stale(LIBTRICO, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIBTRICO] + objs + ["-ldl"])
    if force or _stale(LIBTRICO_A, objs):
        if os.path.exists(LIBTRICO_A):
            os.remove(LIBTRICO_A)
        _run(["ar", "rcs", LIBTRICO_A] + objs)
    if test_hooks:
        os.makedirs(HOOKDIR, exist_ok=True)
        hobjs = [hook_objs.get(o, o) for o in objs]
        if force or _stale(LIBTRICO_HOOKS, hobjs):
            _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIBTRICO_HOOKS] + hobjs + ["-ldl"])
    stale_hooks = os.path.join(LIBDIR, "libtrico_testhooks.so")       # where rounds 2 and 3 put it
    if os.path.exists(stale_hooks):
        os.remove(stale_hooks)
    mg = os.path.join(CSRC, "tools", "meshgen.c")
    if force or _stale(LIBMESHGEN, [mg]):
        _run([CC, "-O2", "-std=c11", "-fPIC", "-fvisibility=hidden", "-shared", mg, "-o", LIBMESHGEN])
    # STL / PLY readers and the two command line tools around the path
    io_src = [os.path.join(CSRC, "io", f) for f in ("iostl.c", "ioply.c")]
    io_hdr = [os.path.join(INCLUDE, "trico_io", h) for h in ("iostl.h", "ioply.h", "trico_io_api.h")]
    if force or _stale(LIBIO, io_src + io_hdr + [LIBTRICO]):
        # links libtrico.so: the STL reader hands big welds to the device (trico_hip_weld_vertices)
        _run([CC, "-O2", "-std=c11", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra", "-shared",
              "-I" + INCLUDE] + io_src + ["-o", LIBIO, "-L" + LIBDIR, "-ltrico", "-Wl,-rpath,$ORIGIN"])
    os.makedirs(BINDIR, exist_ok=True)
    for tool in TOOLS:
        src = os.path.join(CSRC, "tools", tool + ".c")
        exe = os.path.join(BINDIR, tool)
        if force or _stale(exe, [src, LIBIO, LIBTRICO] + io_hdr + headers[:1]):
            # -ffp-contract=off: the decoder's flat normals must round like the reference build
            _run([CC, "-O2", "-std=gnu11", "-ffp-contract=off", "-Wall", "-Wextra", "-I" + INCLUDE, src, "-o", exe,
                  "-L" + LIBDIR, "-ltrico_io", "-ltrico", "-lm", "-Wl,-rpath,$ORIGIN/../lib"])
    return LIBTRICO


if __name__ == "__main__":
    build(force="--force" in sys.argv, test_hooks="--test-hooks" in sys.argv)
