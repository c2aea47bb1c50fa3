"""Build libhmx.so (HIP kernels + C ABI) for gfx950 with plain hipcc, in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container and the
resulting .so travels to the GPU box next to the sources.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhmx.so")
SOURCES = ["hmx_kernels.hip", "hmx_sweep.hip", "hmx_lisi.hip", "hmx_capi.cpp"]
HEADERS = [os.path.join(CSRC, "hmx_internal.h"), os.path.join(CSRC, "hmx_device.h"), os.path.join(ROOT, "include", "hmx.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str | None = None) -> str:
    """``out`` builds a variant library (timing experiments: e.g. -DHMX_ABL=..) next to libhmx.so."""
    if out is not None:
        return _compile(out, verbose, extra_flags)
    if not force and not needs_build():
        return LIB
    return _compile(LIB, verbose, extra_flags)


def _compile(LIB, verbose, extra_flags):
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip"]
    cmd += list(extra_flags)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB + ".tmp"]
    if verbose:
        print("[harmonypy_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = None
    if "-o" in args:
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    print(build(force="--force" in sys.argv, extra_flags=args, out=out))
