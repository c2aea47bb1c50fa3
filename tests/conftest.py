import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Z_corr acceptance (BASELINE.json north_star / SURVEY.md §8c-3): relative
# Frobenius error AND max-abs error relative to max|Z_ref|, both <= 1e-4 (fp32).
Z_TOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def z_errors(Z, Z_ref):
    Z = np.asarray(Z, dtype=np.float64)
    Z_ref = np.asarray(Z_ref, dtype=np.float64)
    rel_f = np.linalg.norm(Z - Z_ref) / np.linalg.norm(Z_ref)
    max_rel = np.abs(Z - Z_ref).max() / np.abs(Z_ref).max()
    return rel_f, max_rel


def assert_z_close(Z, Z_ref, tol=Z_TOL, what="Z_corr"):
    rel_f, max_rel = z_errors(Z, Z_ref)
    assert rel_f <= tol and max_rel <= tol, f"{what}: relF={rel_f:.3e} max={max_rel:.3e} > {tol:g}"
    return rel_f, max_rel


def load_case(name):
    """Golden case -> (data N x d, meta DataFrame, vars_use, kwargs, golden dict)."""
    import pandas as pd
    g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    kwargs = json.loads(str(g["kwargs"]))
    vars_use = json.loads(str(g["vars_use"]))
    if name.startswith("pbmc"):
        inp = np.load(os.path.join(GOLDEN, "pbmc_3500_inputs.npz"))
        data = inp["pcs"]
        meta = pd.DataFrame({"donor": inp["donor"].astype(str), "tech": inp["tech"].astype(str)})
    else:
        inp = np.load(os.path.join(GOLDEN, "synth_small_inputs.npz"))
        data = inp["Z"]
        meta = pd.DataFrame({"batch": np.array([f"b{i}" for i in inp["batch"]])})
    return data, meta, vars_use, kwargs, g


ALL_CASES = ["pbmc_default", "pbmc_seed7", "pbmc_short", "pbmc_fixed_schedule", "pbmc_lambda_est",
             "pbmc_theta_tau", "pbmc_two_vars", "synth_small_steps", "synth_small_default",
             "synth_small_lambda_est"]


def thin_margin_iteration(g, tol=0.05):
    """Index of the first Harmony iteration holding a type-0 decision whose ratio is within
    ``tol`` of the threshold (harmony.py:523), or None."""
    rounds = [int(r) for r in g["kmeans_rounds"]]
    m = list(g["margins"])
    pos = 0
    for it, r in enumerate(rounds):
        n = max(0, r - 4)
        if any(abs(x - 1.0) < tol for x in m[pos:pos + n]):
            return it
        pos += n
    return None


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
