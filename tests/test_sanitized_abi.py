"""ASan + UBSan build of the C-ABI host code, exercised without a GPU: the C99 client and the ABI error
paths run against a library compiled with -fsanitize=address,undefined (host code only: -fno-gpu-sanitize).

The sanitized library (build/libhmx_asan.so) is rebuilt whenever its stamp does not carry the build id of the current
sources (`_build.build_sanitized`: only hmx_capi.cpp is recompiled, the kernel objects are shared with libhmx.so) -- a
stale library is never tested, a missing compiler fails the test instead of skipping it."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

ASAN_LIB = os.path.join(ROOT, "build", "libhmx_asan.so")


def _runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.fixture(scope="module")
def asan_lib():
    from harmonypy_amd import _build
    _build.build(verbose=False)                 # the kernel objects the sanitized library shares
    lib = _build.build_sanitized(ASAN_LIB)
    with open(lib + ".buildid") as f:
        assert f.read().strip() == _build.build_id(_build.ASAN_FLAGS), "sanitized library is stale"
    assert _runtime() is not None, "clang's ASan runtime not found under /opt/rocm/lib/llvm"
    return lib


def _env():
    return dict(os.environ, LD_PRELOAD=_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_c_client_under_asan(asan_lib, tmp_path):
    """tests/c/abi_check.c (dlopen, every symbol, hmx_create error paths) against the sanitized library."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "abi_check")
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_check.c"),
                    "-ldl", "-o", exe], check=True)
    r = subprocess.run([exe, asan_lib], capture_output=True, text=True, env=_env())
    assert r.returncode == 0 and r.stdout.startswith("ok abi="), r.stdout + r.stderr


def test_abi_error_paths_under_asan(asan_lib):
    """Argument and state errors of the ABI through ctypes in a sanitized child process: null pointers, sizes out of
    range, calls before hmx_create succeeded (no GPU here: creation itself fails with a message, never a crash)."""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r)
os.environ["HMX_LIB"] = %r
from harmonypy_amd import _capi
lib = _capi.load()
assert lib.hmx_abi_version() == _capi.HMX_ABI_VERSION
cfg = _capi.HmxConfig(n_cells=64, n_cells_global=0, n_pcs=5, n_clusters=3, n_batches=2, n_groups=2, n_vars=1, n_blocks=20, device_id=0)
h = C.c_void_p()
rc = lib.hmx_create(C.byref(cfg), C.byref(h))
assert rc != 0 and h.value is None and lib.hmx_last_error()
for bad in (dict(n_clusters=999), dict(n_pcs=0), dict(n_blocks=1000), dict(n_vars=33), dict(n_cells=-1)):
    c2 = _capi.HmxConfig(n_cells=64, n_cells_global=0, n_pcs=5, n_clusters=3, n_batches=2, n_groups=2, n_vars=1, n_blocks=20, device_id=0)
    for k, v in bad.items(): setattr(c2, k, v)
    assert lib.hmx_create(C.byref(c2), C.byref(h)) == -1
assert lib.hmx_create(None, C.byref(h)) == -1
import numpy as np
out = np.zeros(16, np.int64)
assert lib.hmx_counters(None, out.ctypes.data_as(C.c_void_p)) == -1
assert lib.hmx_sync(None) == -1 and lib.hmx_moe_correct_ridge(None) == -1
lib.hmx_destroy(None)
print("sanitized abi ok")
''' % (ROOT, asan_lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=_env())
    assert r.returncode == 0 and "sanitized abi ok" in r.stdout, r.stdout + r.stderr[-3000:]
