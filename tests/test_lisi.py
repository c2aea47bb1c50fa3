"""LISI: the oracle restatement against the reference's known answers (CPU), the device against both (GPU)."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import lisi_oracle as LO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    return g["X"], g["codes"], g["lisi_reference"], int(g["perplexity"]), g


def _meta(codes):
    return pd.DataFrame({f"l{i}": pd.Categorical(c) for i, c in enumerate(codes)})


def test_oracle_reproduces_reference_known_answer():
    """The fixture of the reference's tests/test_lisi.py:5-17 (np.allclose, its tolerance)."""
    X, codes, ref, perp, g = _load("lisi_ref_fixture")
    out = LO.compute_lisi(X, list(codes), perp)
    assert np.allclose(out, g["lisi_stored"])
    np.testing.assert_allclose(out, ref, rtol=1e-12)


def test_oracle_matches_reference_on_pbmc_subset():
    X, codes, ref, perp, _ = _load("lisi_pbmc_p30")
    pick = np.arange(0, X.shape[0], 7)
    dist, idx = LO.knn_exact(X, perp * 3)
    out = np.array([[1.0 / LO.simpson_cell(dist[i, 1:], c[idx[i, 1:]], perp) for c in codes] for i in pick])
    np.testing.assert_allclose(out, ref[pick], rtol=1e-10)


def test_oracle_degenerate_entropy():
    # neighbours so far away that exp(-D) underflows at every beta tried first: H == 0 -> simpson -1 (lisi.py:121-122)
    D = np.full(5, 1e6)
    assert LO.simpson_cell(D, np.zeros(5, int), 30) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lisi_ref_fixture", "lisi_pbmc_p30", "lisi_pbmc_p15"])
def test_device_lisi_matches_reference(name):
    import harmonypy_amd as hm
    X, codes, ref, perp, g = _load(name)
    meta = _meta(codes)
    out = hm.compute_lisi(X, meta, meta.columns, perp)
    assert out.shape == ref.shape and out.dtype == np.float64
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=0)
    if "lisi_stored" in g:
        assert np.allclose(out, g["lisi_stored"])                     # the reference's own test criterion


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,perp,seed", [(1000, 2, 30, 0), (5000, 50, 30, 1), (777, 20, 10, 2), (4099, 64, 40, 3),
                                          (1500, 100, 30, 4), (1200, 200, 20, 5), (91, 3, 30, 6),
                                          # the larger candidate lists: 1024 entries (up to 504 neighbours), 4096 (up to 2040)
                                          (3000, 50, 41, 7), (2500, 20, 100, 8), (1800, 8, 168, 9), (2600, 50, 200, 10),
                                          (2100, 16, 680, 11)])
def test_device_neighbours_are_exact(n, d, perp, seed):
    """Neighbour sets, order and distances against the float64 brute-force search of the oracle."""
    import harmonypy_amd as hm
    rng = np.random.default_rng(seed)
    cent = rng.normal(size=(8, d)) * 3
    X = cent[rng.integers(0, 8, n)] + rng.normal(size=(n, d)) + 50.0     # off-centre on purpose
    lab = rng.integers(0, 4, n)
    meta = pd.DataFrame({"a": lab.astype(str)})
    out, kd, ki = hm.compute_lisi(X, meta, ["a"], perp, return_neighbors=True)
    dist, idx = LO.knn_exact(X, perp * 3)
    np.testing.assert_array_equal(ki, idx[:, 1:])
    np.testing.assert_allclose(kd, dist[:, 1:], rtol=1e-12)
    want = np.array([1.0 / LO.simpson_cell(dist[i, 1:], lab[idx[i, 1:]], perp) for i in range(0, n, 13)])
    np.testing.assert_allclose(out[::13, 0], want, rtol=1e-9)


@pytest.mark.gpu
def test_device_lisi_errors():
    import harmonypy_amd as hm
    X = np.random.default_rng(0).normal(size=(50, 3))
    meta = pd.DataFrame({"a": ["x"] * 50})
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta, ["a"], 30)                           # 90 neighbours of 50 points (sklearn's ValueError)
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta, ["a"], 60)                           # 180 neighbours of 50 points
    with pytest.raises(ValueError):
        hm.compute_lisi(X, meta, ["a"], 5, device="cpu")
    assert hm.compute_lisi(X, meta, ["a"], 5).shape == (50, 1)
    np.testing.assert_allclose(hm.compute_lisi(X, meta, ["a"], 5), 1.0)  # one category: LISI = 1


def test_perplexity_beyond_the_build_limit_is_a_value_error():
    """3 * perplexity > 2040 neighbours (the largest candidate list): refused with ValueError before any device work
    (lisi.py:53 takes any)."""
    import pandas as pd
    import harmonypy_amd
    X = np.random.default_rng(0).normal(size=(500, 5))
    meta = pd.DataFrame({"b": np.arange(500) % 3})
    with pytest.raises(ValueError, match="perplexity"):
        harmonypy_amd.compute_lisi(X, meta, ["b"], perplexity=681)
