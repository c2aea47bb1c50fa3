"""k-means++ seeding: the oracle restatement against sklearn (CPU), the device against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import kmeans_seed as ks


def _points(n, d, n_types, seed):
    rng = np.random.default_rng(seed)
    cent = rng.normal(size=(n_types, d)) * 2
    X = (cent[rng.integers(0, n_types, n)] + rng.normal(size=(n, d))).astype(np.float32)
    return X / np.linalg.norm(X, axis=1, keepdims=True)


def _potential(X, C):
    d2 = (X * X).sum(1)[:, None] - 2 * X @ C.T + (C * C).sum(1)[None]
    return float(np.maximum(d2, 0).min(1).sum())


def test_oracle_matches_sklearn_statistically():
    """Same algorithm, different generator: the seeding potentials agree in distribution."""
    from sklearn.cluster import kmeans_plusplus
    X = _points(6000, 30, 40, 0)
    sk = [_potential(X, kmeans_plusplus(X, n_clusters=50, random_state=s)[0]) for s in range(8)]
    ours = [_potential(X, ks.kmeans_plusplus(X, 50, seed=s)[0]) for s in range(8)]
    assert abs(np.mean(ours) - np.mean(sk)) < 0.04 * np.mean(sk)
    # and both are far better than uniform seeding
    rng = np.random.default_rng(1)
    uni = np.mean([_potential(X, X[rng.choice(len(X), 50, replace=False)]) for _ in range(8)])
    assert np.mean(ours) < 0.9 * uni


def test_oracle_is_deterministic_and_picks_points():
    X = _points(3000, 20, 10, 2)
    C1, ch1 = ks.kmeans_plusplus(X, 30, seed=7)
    C2, ch2 = ks.kmeans_plusplus(X, 30, seed=7)
    assert np.array_equal(ch1, ch2) and np.array_equal(C1, C2)
    assert len(set(ch1.tolist())) == 30 and np.array_equal(C1, X[ch1])
    assert not np.array_equal(ch1, ks.kmeans_plusplus(X, 30, seed=8)[1])


def test_oracle_generator_known_answers():
    # splitmix64 finaliser: fixed points of the restated generator (guards against silent edits)
    assert ks.rand64(0, 0, 0) == 0xE220A8397B1DCDAF          # splitmix64(seed=0) first output
    assert ks.rand64(0, 0, 1) == 0x6E789E6AA1B965F4          # second output


def test_oracle_degenerate_points():
    X = np.tile(_points(1, 8, 1, 0), (50, 1))                # every point identical: potential 0
    C, ch = ks.kmeans_plusplus(X, 5, seed=1)
    assert C.shape == (5, 8) and np.all((ch >= 0) & (ch < 50))


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,K,seed", [(5000, 30, 100, 0), (32768, 50, 100, 1), (300, 12, 20, 2), (70001, 20, 30, 3), (64, 5, 8, 4)])
def test_device_seeding_bit_exact(n, d, K, seed):
    from harmonypy_amd import _capi
    X = _points(n, d, max(K // 2, 3), seed)
    eng = _capi.Engine(16, d, K, 1, 1, 1, 20)                # seeding needs no upload
    C, ch = eng.kmeans_seed(X, seed)
    Co, cho = ks.kmeans_plusplus(X, K, seed)
    assert np.array_equal(ch, cho)
    assert np.array_equal(C, Co)
    eng.close()


@pytest.mark.gpu
def test_device_seeding_degenerate():
    from harmonypy_amd import _capi
    X = np.tile(_points(1, 8, 1, 0), (700, 1))
    eng = _capi.Engine(16, 8, 5, 1, 1, 1, 20)
    C, ch = eng.kmeans_seed(X, 1)
    Co, cho = ks.kmeans_plusplus(X, 5, 1)
    assert np.array_equal(ch, cho) and np.array_equal(C, Co)
    eng.close()
