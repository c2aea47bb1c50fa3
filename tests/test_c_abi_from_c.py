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


@pytest.mark.gpu
def test_c_client_computes_end_to_end(tmp_path, monkeypatch):
    """tests/c/e2e_client.c -- create, upload, init_cluster, hmx_cluster, moe_correct_ridge, hmx_get from plain C, with the
    host side of the boundary (group-sorted layout, static tiles, Pr_b, theta / lamb / sigma, blocks) restated in C -- on a
    2 000-cell case: Z_corr within 1e-4 of the oracle run on the same update order (harmony.py:437-462, 535-569), within
    2e-6 of the Python binding driving the same calls, the objective terms of every round equal to the binding's."""
    import sys
    import numpy as np
    import pandas as pd
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    from conftest import assert_z_close
    from harmonypy_amd import _capi
    from harmonypy_amd import harmony as H
    from oracle.device_order import positions
    from oracle.harmony_oracle import OracleHarmony, prepare_inputs
    _capi.load()
    exe = str(tmp_path / "e2e_client")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "e2e_client.c"), "-ldl", "-lm", "-o", exe], check=True)
    rng = np.random.default_rng(21)
    N, d, K, B, rounds, seed = 2000, 20, 10, 3, 6, 77
    batch = rng.integers(0, B, size=N).astype(np.int32)
    centres = rng.normal(0, 2.0, (5, d))
    Z = (centres[rng.integers(0, 5, size=N)] + rng.normal(size=(N, d)) + batch[:, None] * 0.7).astype(np.float32)
    Zn = Z / np.linalg.norm(Z, axis=1, keepdims=True)
    Y0 = Zn[rng.choice(N, K, replace=False)].astype(np.float32)          # K x d: any K cells as starting centroids
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([N, d, K, B, rounds, seed], np.int32).tobytes())
        f.write(np.ascontiguousarray(Z).tobytes())
        f.write(batch.tobytes())
        f.write(np.ascontiguousarray(Y0).tobytes())
    r = subprocess.run([exe, _capi.LIB_PATH, str(inp), str(outp)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("e2e ok"), r.stdout + r.stderr
    print(r.stdout.strip())
    raw = open(outp, "rb").read()
    hdr = np.frombuffer(raw[:12], np.int32)
    assert list(hdr) == [N, d, rounds]
    terms = np.frombuffer(raw[12:12 + rounds * 32], np.float64).reshape(rounds, 4)
    Zc = np.frombuffer(raw[12 + rounds * 32:], np.float32).reshape(N, d)
    # the oracle on the same update order
    meta = pd.DataFrame({"batch": [f"b{i}" for i in batch]})
    p = prepare_inputs(Z, meta, ["batch"], nclust=K)
    state = {"counter": 0}

    def perm(n):
        pos = positions(np.arange(N), N, seed, state["counter"])
        state["counter"] += 1
        return np.argsort(pos, kind="stable")
    oo = OracleHarmony(p["Z"], p["phi"], p["Pr_b"], p["sigma"], p["theta"], p["lamb"], K=K, run=False, perm_source=perm,
                       forced_rounds=[rounds])
    oo.init_cluster(seed, Y0.T)
    oo.cluster()
    oo.moe_correct_ridge()
    rel_f, max_rel = assert_z_close(Zc, oo.result(), what="C client vs oracle")
    obj_c = terms[:, :3].sum(axis=1) * 2000.0 / N
    np.testing.assert_allclose(obj_c, oo.objective_kmeans[1:], rtol=2e-5)
    # the Python binding driving the same calls
    monkeypatch.setenv("HMX_UPDATE_ORDER", "device")
    ho = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, random_state=seed, verbose=False, _y0=Y0.T)
    ho.cluster(_rounds=rounds)
    ho.moe_correct_ridge()
    rel_p, max_p = assert_z_close(Zc, ho.Z_corr, tol=2e-6, what="C client vs Python binding")
    np.testing.assert_allclose(obj_c, ho.objective_kmeans[1:], rtol=1e-6)
    print(f"C client: Z_corr vs oracle relF={rel_f:.2e} max={max_rel:.2e}; vs Python binding relF={rel_p:.1e}")
