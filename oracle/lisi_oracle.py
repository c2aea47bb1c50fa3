"""CPU restatement of the reference's LISI (harmonypy/lisi.py) -- TEST INFRASTRUCTURE ONLY.

``compute_lisi`` (lisi.py:24-66): exact Euclidean k-nearest neighbours with ``n_neighbors = 3 *
perplexity`` (the reference asks sklearn's kd-tree, third-party; restated here as a brute-force search
in float64, distances of the selected neighbours from direct differences like the tree's), the first
column dropped ("don't count yourself", lisi.py:58-60), then per label column ``1 / compute_simpson``.
``compute_simpson`` (lisi.py:69-133): per cell a bisection on beta until the entropy of
``P = exp(-beta * D)`` matches ``log(perplexity)`` within ``tol`` (at most 50 tries), then the sum over
categories of the squared neighbourhood probability mass; -1 when the entropy is exactly 0.

Pinned by tests/test_lisi.py against the reference's own known-answer files (data/lisi_*.tsv.gz, the
fixture of the reference's tests/test_lisi.py:5-17) and against the reference run on pbmc_3500
(tests/golden/lisi_*.npz, made by tests/golden/make_lisi_golden.py).  Only tests/ may import it.
"""
import numpy as np


def knn_exact(X, n_neighbors, chunk=2048):
    """Sorted distances and indices (n x n_neighbors), the query itself included, ties by index."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    if n_neighbors > n:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors}, n_samples_fit = {n}")
    sq = (X * X).sum(1)
    dist = np.empty((n, n_neighbors))
    idx = np.empty((n, n_neighbors), dtype=np.int64)
    pool = min(n, n_neighbors + 16)                # a few more than needed: the ranking below is exact
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        d2 = sq[lo:hi, None] - 2.0 * (X[lo:hi] @ X.T) + sq[None, :]
        cand = np.argpartition(d2, pool - 1, axis=1)[:, :pool]
        for r in range(hi - lo):
            c = cand[r]
            diff = X[c] - X[lo + r]
            dd = np.sqrt((diff * diff).sum(1))     # as the kd-tree reports them
            o = np.lexsort((c, dd))[:n_neighbors]
            dist[lo + r] = dd[o]
            idx[lo + r] = c[o]
    return dist, idx


def simpson_cell(D, lab, perplexity, tol=1e-5):
    """lisi.py:83-132 for one cell: D distances to its neighbours, lab their label codes."""
    logU = np.log(perplexity)
    beta, betamin, betamax = 1.0, -np.inf, np.inf

    def entropy(beta):
        P = np.exp(-D * beta)
        P_sum = np.sum(P)
        if P_sum == 0:
            return 0.0, np.zeros(D.shape[0])
        H = np.log(P_sum) + beta * np.sum(D * P) / P_sum
        return H, P / P_sum

    H, P = entropy(beta)
    Hdiff = H - logU
    for _ in range(50):
        if abs(Hdiff) < tol:
            break
        if Hdiff > 0:
            betamin = beta
            beta = beta * 2 if not np.isfinite(betamax) else (beta + betamax) / 2
        else:
            betamax = beta
            beta = beta / 2 if not np.isfinite(betamin) else (beta + betamin) / 2
        H, P = entropy(beta)
        Hdiff = H - logU
    s = -1.0 if H == 0 else 0.0
    for c in np.unique(lab):
        m = np.sum(P[lab == c])
        s += m * m
    return s


def compute_lisi(X, label_codes, perplexity=30):
    """X: n x d; label_codes: list of integer code arrays (one per label column).  Returns n x n_labels."""
    n_neighbors = int(perplexity * 3)
    dist, idx = knn_exact(X, n_neighbors)
    dist, idx = dist[:, 1:], idx[:, 1:]            # lisi.py:58-60
    out = np.zeros((X.shape[0], len(label_codes)))
    for j, codes in enumerate(label_codes):
        codes = np.asarray(codes)
        for i in range(X.shape[0]):
            out[i, j] = 1.0 / simpson_cell(dist[i], codes[idx[i]], perplexity)
    return out
