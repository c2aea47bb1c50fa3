"""Build libhmx.so (HIP kernels + C ABI) for gfx950 with plain hipcc, in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container and the
resulting .so travels to the GPU box next to the sources.

Every source is compiled to its own object (in parallel; objects are cached under ``build/`` by a hash
of the source, the headers and the flags) and the objects are linked.  The library carries a BUILD ID =
the first 12 hex digits of the SHA-256 over csrc/* and include/hmx.h (``hmx_build_id()``): profiles and
counter files name the build they were collected on, and ``bench.py`` only quotes counter-derived traffic
for the build that is running -- no hand-maintained version string.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhmx.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
SOURCES = ["hmx_kernels.hip", "hmx_rtz3.hip", "hmx_lisi.hip", "hmx_capi.cpp"]
HEADERS = [os.path.join(CSRC, "hmx_internal.h"), os.path.join(CSRC, "hmx_device.h"), os.path.join(ROOT, "include", "hmx.h")]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources(extra_flags=()) -> list:
    return list(SOURCES)


def build_id(extra_flags=()) -> str:
    """Hash of everything that decides the kernels: csrc/*, include/hmx.h, the flags."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    for p in files + [os.path.join(ROOT, "include", "hmx.h")]:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    flags = sorted(set(extra_flags))
    h.update(" ".join(BASE_FLAGS + flags).encode())
    return h.hexdigest()[:12]


def _stamp_path(lib):
    return lib + ".buildid"


def needs_build(extra_flags=()) -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        with open(_stamp_path(LIB)) as f:
            return f.read().strip() != build_id(extra_flags)
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str | None = None) -> str:
    """``out`` builds a variant library (timing experiments: e.g. -DHMX_ROUND_PROF) next to libhmx.so."""
    extra_flags = list(extra_flags)
    if out is not None:
        return _compile(out, verbose, extra_flags)
    if not force and not needs_build(extra_flags):
        return LIB
    return _compile(LIB, verbose, extra_flags)


ASAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer"]


def build_sanitized(out: str, verbose: bool = False) -> str:
    """ASan + UBSan build of the C-ABI HOST code: hmx_capi.cpp is compiled with the sanitizers, the kernel files keep
    their ordinary objects (their host side is launch glue) -- seconds when libhmx.so was just built.  Rebuilt whenever
    the stamp next to ``out`` does not carry the current build id."""
    want = build_id(ASAN_FLAGS)
    try:
        with open(_stamp_path(out)) as f:
            if f.read().strip() == want and os.path.exists(out):
                return out
    except OSError:
        pass
    os.makedirs(os.path.dirname(out), exist_ok=True)
    return _compile(out, verbose, [], capi_flags=ASAN_FLAGS, stamp=want)


def _object(src, flags, verbose):
    """Compile one source to a cached object; the cache key covers the source, the headers and the flags."""
    path = os.path.join(CSRC, src)
    h = hashlib.sha256()
    for p in [path] + HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    obj = os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{h.hexdigest()[:16]}.o")
    if os.path.exists(obj):
        return obj
    cmd = [_hipcc()] + flags + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip", "-c", path, "-o", obj + ".tmp"]
    if verbose:
        print("[harmonypy_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp", obj)
    # keep the cache small: the four newest objects per source
    stem = os.path.splitext(src)[0] + "."
    old = sorted((f for f in os.listdir(OBJ_DIR) if f.startswith(stem) and f.endswith(".o")),
                 key=lambda f: os.path.getmtime(os.path.join(OBJ_DIR, f)))
    for f in old[:-4]:
        try:
            os.remove(os.path.join(OBJ_DIR, f))
        except OSError:
            pass
    return obj


def _compile(lib, verbose, extra_flags, capi_flags=(), stamp=None):
    os.makedirs(OBJ_DIR, exist_ok=True)
    bid = stamp or build_id(extra_flags)
    flags = BASE_FLAGS + list(extra_flags)
    srcs = sources(extra_flags)

    def one(src):
        f = list(flags)
        if src == "hmx_capi.cpp":
            f += list(capi_flags) + [f'-DHMX_BUILD_ID="{bid}"']
        return _object(src, f, verbose)
    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(one, srcs))
    link_flags = [f for f in list(extra_flags) + list(capi_flags) if f.startswith("-fsanitize") or f == "-fno-gpu-sanitize"]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + link_flags + objs + ["-o", lib + ".tmp"]
    if verbose:
        print("[harmonypy_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    with open(_stamp_path(lib), "w") as f:
        f.write(bid + "\n")
    return lib


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    print(build(force="--force" in sys.argv, extra_flags=args, out=out))
