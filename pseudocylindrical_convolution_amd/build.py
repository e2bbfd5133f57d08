"""Build the native libraries in-tree.

  libpconv_hip.so    HIP kernels + host geometry, gfx950 only (hipcc)
  libpconv_coder.so  CPU arithmetic coder (g++, no GPU code)

Both land next to this file so that they travel with the source tree.  Objects
are rebuilt only when a source or header is newer.  `python -m
pseudocylindrical_convolution_amd.build` builds everything.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
INCLUDE = os.path.join(ROOT, "include")

HIP_SOURCES = [
    "geometry.cpp",
    "resample.hip",
    "tilepad.hip",
    "pointwise.hip",
    "entropy.hip",
    "entropy_engine.hip",
    "entropy_mfma.hip",
    "conv.hip",
    "wino.hip",
    "wino_flat.hip",   # wino.hip again with a 2-row x 128-column workgroup tile (the row split's remainders)
    "wino42.hip",
    "backward.hip",
    "engine.cpp",
    "coder.cpp",  # the engine drives the arithmetic coder natively
]
CODER_SOURCES = ["coder.cpp"]

ARCH = "gfx950"
# -ffp-contract=off: the gather/lerp/CDF kernels are specified operation by
# operation (parity with the oracle is bit-exact); fused multiply-adds are
# written explicitly (fmaf / MFMA) where they are part of the contract.
HIP_FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
    "-I" + INCLUDE,
]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-I" + INCLUDE]
FILE_FLAGS = {}  # per-file additions to HIP_FLAGS (none at present)
INCLUDES = {"wino_flat.hip": ["wino.hip"]}  # sources that #include another source: rebuilt when it changes


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpconv_hip.so cannot be built")


def _newest_header():
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _stale(target, deps_mtime):
    return (not os.path.exists(target)) or os.path.getmtime(target) < deps_mtime


def _older_than(lib, objs):
    """a library is relinked when it is missing or older than any of its own objects"""
    return (not os.path.exists(lib)) or any(os.path.getmtime(lib) < os.path.getmtime(o) for o in objs)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    if r.stdout.strip():
        sys.stderr.write(r.stdout)


def _compile(src, flags, cc, hdr_time, as_hip):
    # one object directory per toolchain: coder.cpp is compiled twice (hipcc for the
    # engine inside libpconv_hip.so, g++ for libpconv_coder.so) and the two objects
    # must never stand in for each other
    path = os.path.join(CSRC, src)
    objdir = os.path.join(OBJ, "hip" if as_hip else "cxx")
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, src + ".o")
    dep_time = max([os.path.getmtime(path)] + [os.path.getmtime(os.path.join(CSRC, d)) for d in INCLUDES.get(src, [])])
    if _stale(obj, max(dep_time, hdr_time)):
        cmd = [cc] + flags + (FILE_FLAGS.get(src, []) if as_hip else [])
        if as_hip and src.endswith(".cpp"):
            cmd += ["-x", "hip"]
        _run(cmd + ["-c", path, "-o", obj])
        return obj, True
    return obj, False


def build(verbose=False, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdr = _newest_header()
    extra = os.environ.get("PCONV_EXTRA_HIPFLAGS", "").split()  # tuning experiments only
    if extra:
        HIP_FLAGS.extend(extra)
        hdr = float("inf")  # force a rebuild with the extra flags
    sources = [s for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, HIP_FLAGS, cc, hdr, True), sources))
    objs = [o for o, _ in res]
    lib = os.path.join(HERE, "libpconv_hip.so")
    if any(ch for _, ch in res) or _older_than(lib, objs):
        _run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs)
        if verbose:
            print("linked", lib)
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler found for libpconv_coder.so")
    cres = [_compile(s, CXX_FLAGS, cxx, hdr, False) for s in CODER_SOURCES]
    clib = os.path.join(HERE, "libpconv_coder.so")
    if any(ch for _, ch in cres) or _older_than(clib, [o for o, _ in cres]):
        _run([cxx, "-shared", "-fPIC", "-o", clib] + [o for o, _ in cres])
        if verbose:
            print("linked", clib)
    return lib, clib


if __name__ == "__main__":
    print(build(verbose=True))
