"""The oracle (oracle/harmony_oracle.py) against fixtures generated from the reference.

CPU only.  Pins the NumPy restatement to the reference's own outputs
(tests/golden/make_golden.py) before anything else is allowed to trust it.
"""
import numpy as np
import pytest

from conftest import ALL_CASES, assert_z_close, load_case, thin_margin_iteration
from oracle import oracle_run_harmony


def test_randperm_stream_matches_fixture():
    """The block order is torch's CPU randperm stream (harmony.py:200,471)."""
    import torch
    _, meta, _, kw, g = load_case("pbmc_default")
    torch.manual_seed(kw.get("random_state", 0))
    perm = torch.randperm(meta.shape[0]).numpy()
    assert np.array_equal(perm[:64], g["first_perm_head"])
    crc = int(np.bitwise_xor.reduce(perm * np.arange(1, len(perm) + 1)))
    assert crc == int(g["first_perm_crc"][0])


@pytest.mark.parametrize("case", ALL_CASES)
def test_oracle_forced_schedule(case):
    """Same Y0, same permutations, the reference's round schedule replayed."""
    data, meta, vars_use, kw, g = load_case(case)
    oo = oracle_run_harmony(data, meta, vars_use, Y0=g["Y0"],
                            forced_rounds=[int(r) for r in g["kmeans_rounds"]],
                            **{k: v for k, v in kw.items()})
    # with a forced schedule the outer loop may still stop early only where the reference did
    n_it = len(g["kmeans_rounds"])
    assert oo.kmeans_rounds == [int(r) for r in g["kmeans_rounds"]][:len(oo.kmeans_rounds)]
    assert len(oo.kmeans_rounds) == n_it
    assert_z_close(oo.result(), g["Z_corr"])
    np.testing.assert_allclose(oo.objective_harmony, g["objective_harmony"], rtol=2e-5)
    np.testing.assert_allclose(oo.objective_kmeans, g["objective_kmeans"], rtol=2e-5)
    np.testing.assert_allclose(oo.O, g["O"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(oo.E, g["E"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ALL_CASES)
def test_oracle_natural_run(case):
    """Free-running convergence: equal schedule => parity; a different schedule is only
    acceptable where the reference's own decision sat within 5 % of the threshold."""
    data, meta, vars_use, kw, g = load_case(case)
    oo = oracle_run_harmony(data, meta, vars_use, Y0=g["Y0"], **kw)
    ref_rounds = [int(r) for r in g["kmeans_rounds"]]
    if oo.kmeans_rounds == ref_rounds:
        assert_z_close(oo.result(), g["Z_corr"])
        return
    first_bad = next(i for i, (a, b) in enumerate(zip(oo.kmeans_rounds + [None] * 99, ref_rounds + [None] * 99))
                     if a != b)
    thin = thin_margin_iteration(g)
    assert thin is not None and thin <= first_bad, (
        f"{case}: schedule {oo.kmeans_rounds} != {ref_rounds} although no decision was marginal")


def test_oracle_step_level():
    """R / O / E / Y after init and after every update_R, Z after every ridge."""
    data, meta, vars_use, kw, g = load_case("synth_small_steps")
    got = []
    hooks = {
        "init_cluster": lambda s: got.append(("init_cluster", dict(R=s.R.T.copy(), O=s.O.copy(), E=s.E.copy(), Y=s.Y.copy()))),
        "update_R": lambda s: got.append(("update_R", dict(R=s.R.T.copy(), O=s.O.copy(), E=s.E.copy(), Y=s.Y.copy()))),
        "ridge": lambda s: got.append(("moe_correct_ridge", dict(Z_corr=s.Z_corr.T.copy(), Z_cos=s.Z_cos.T.copy()))),
    }
    oracle_run_harmony(data, meta, vars_use, Y0=g["Y0"], hooks=hooks, **kw)
    assert len(got) == int(g["n_steps"][0])
    for i, (name, arrays) in enumerate(got):
        for key, val in arrays.items():
            ref = g[f"step{i:03d}.{name}.{key}"]
            if key in ("Z_corr", "Z_cos"):
                assert_z_close(val, ref, what=f"step {i} {name}.{key}")
            else:
                np.testing.assert_allclose(val, ref, rtol=5e-4, atol=2e-5, err_msg=f"step {i} {name}.{key}")


def test_reference_own_acceptance_on_oracle():
    """The reference's CI test: per-PC Pearson r >= 0.9 against the R package output
    (tests/test_harmony.py:24-30,130), here applied to the oracle."""
    import os
    from scipy.stats import pearsonr
    from conftest import GOLDEN
    data, meta, vars_use, kw, g = load_case("pbmc_default")
    r_golden = np.load(os.path.join(GOLDEN, "pbmc_3500_inputs.npz"))["r_harmonized"]
    oo = oracle_run_harmony(data, meta, vars_use, Y0=g["Y0"], **kw)
    cors = [pearsonr(oo.result()[:, j], r_golden[:, j])[0] for j in range(data.shape[1])]
    assert min(cors) >= 0.9
