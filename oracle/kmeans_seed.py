"""CPU restatement of the engine's k-means++ seeding (hmx_kmeans_seed) -- TEST INFRASTRUCTURE ONLY.

The reference obtains its initial centroids from scikit-learn (third-party, unpinned in the
reference's pyproject.toml; 1.7.2 in this image): ``KMeans(n_clusters=K, init='k-means++', n_init=1,
max_iter=25, random_state=rs).fit(Z_cos.T)`` at harmony.py:370-372.  The seeding algorithm restated
here is sklearn's greedy k-means++ (``sklearn/cluster/_kmeans.py::_kmeans_plusplus``, uniform sample
weights): first centre uniform; then for every further centre ``2 + int(log K)`` candidates drawn
with probability proportional to the squared distance to the closest centre so far
(``searchsorted(cumsum(closest), u * potential)``), the candidate leaving the smallest potential kept.

What differs from sklearn, on purpose: the uniform draws come from a counter-based generator
(splitmix64 of seed/step/trial) instead of NumPy's MT19937 stream, and the sums that decide an index
are 32.32 fixed-point integers so that the device (atomics, any order) and this file agree bit for
bit.  Squared distances are accumulated over the features in order, one float32 rounding per
operation, exactly as the kernel does (k_seed_eval).  Parity with sklearn itself is statistical
(tests/test_kmeans_seed.py compares the potentials of the two on the same points).

Only tests/ may import this module.
"""
import math

import numpy as np

SLOTS = 8                     # candidate slots per step (SEED_TRIALS in hmx_kernels.hip)
_M64 = (1 << 64) - 1


def rand64(seed, step, trial):
    """splitmix64 finaliser of the counter (seed_rand in hmx_kernels.hip)."""
    z = (seed + 0x9E3779B97F4A7C15 * (1 + step * SLOTS + trial)) & _M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M64
    z ^= z >> 31
    return z


def _fx(v):
    """float32 squared distances -> 32.32 fixed point (seed_fx)."""
    return (v.astype(np.float32) * np.float32(4294967296.0)).astype(np.uint64)


def _sqdist(X, c):
    """Squared distances of the rows of X to c: features in order, float32 rounding per operation."""
    acc = np.zeros(X.shape[0], np.float32)
    for j in range(X.shape[1]):
        df = X[:, j] - c[j]
        acc = acc + df * df
    return acc


def kmeans_plusplus(X, K, seed=0):
    """Returns (centres K x d float32, chosen point indices)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n = X.shape[0]
    n_trials = min(SLOTS, 2 + int(math.log(K)))
    seed = int(seed) & _M64
    chosen = [(rand64(seed, 0, 0) * n) >> 64]
    closest = _sqdist(X, X[chosen[0]])
    for step in range(1, K):
        fx = _fx(closest)
        csum = np.cumsum(fx, dtype=np.uint64)
        total = int(csum[-1])
        best = None
        for j in range(n_trials):
            u = rand64(seed, step, j)
            if total == 0:
                cand = (u * n) >> 64
            else:
                target = (u * total) >> 64
                cand = int(np.searchsorted(csum, np.uint64(target), side="right"))
            m = np.minimum(closest, _sqdist(X, X[cand]))
            pot = int(_fx(m).sum(dtype=np.uint64))
            if best is None or pot < best[0]:
                best = (pot, cand, m)
        chosen.append(best[1])
        closest = best[2]
    chosen = np.asarray(chosen, dtype=np.int32)
    return X[chosen].copy(), chosen
