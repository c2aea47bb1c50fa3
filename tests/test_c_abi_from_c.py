"""The boundary is a C ABI: a C99 program includes include/hmx.h, dlopens libhmx.so and calls it (no GPU needed)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_header_is_c99_and_library_serves_a_c_client(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    from harmonypy_amd import _capi
    _capi.load()                                                     # builds nothing: the library must exist
    exe = str(tmp_path / "abi_check")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_check.c"), "-ldl", "-o", exe], check=True)
    out = subprocess.run([exe, _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert out.startswith(f"ok abi={_capi.HMX_ABI_VERSION} sizeof(hmx_config)=")
    # every export the Python binding knows is in the C client's list too
    src = open(os.path.join(ROOT, "tests", "c", "abi_check.c")).read()
    for name in _capi.EXPORTS:
        assert f'"{name}"' in src, name
