"""NumPy fp32 restatement of the harmonypy ``harmonize()`` path (CPU oracle).

TEST INFRASTRUCTURE.  Not shipped, not imported by ``harmonypy_amd``; see
``oracle/__init__.py``.

Pinning: this restatement is pinned against the reference itself.  In the
build container ``tests/golden/make_golden.py`` imports ``/root/reference``
(harmonypy v0.2.0, ``device='cpu'``), runs it on the bundled pbmc_3500 fixture
and on seeded synthetic inputs, and stores step-level and end-to-end outputs
under ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` replays those
cases through this file.  The reference's own tests only pin the path to
Pearson r >= 0.9 against an R-generated file (tests/test_harmony.py:130) and
to same-seed reproducibility (tests/test_harmony.py:57); both are repeated in
our tests as secondary assertions.

Arithmetic lives in third-party torch/sklearn in the reference
(pyproject.toml:25-31, unpinned; build container: torch 2.10.0, sklearn
1.7.2).  This file restates the torch ops with NumPy float32 ops of the same
shape and order; the random block order comes from the very same generator
the reference uses (``torch.manual_seed`` + ``torch.randperm`` on the CPU
generator, harmony.py:200,471) and the initial centroids from the same
sklearn call (harmony.py:370-372).

Layout follows the reference (feature-major: d x N, K x N, B x N) so that
reductions run over the same axes in the same order.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# front end: harmony.py:116-173  (argument normalisation of run_harmony)
# --------------------------------------------------------------------------
def prepare_inputs(data_mat, meta_data, vars_use, theta=None, lamb=None,
                   sigma=0.1, nclust=None, tau=0):
    """Restates harmony.py:116-173.  Returns a dict of plain arrays.

    ``phi`` is the dense one-hot (B x N) exactly as ``pd.get_dummies`` lays it
    out (variables in ``vars_use`` order, levels sorted within a variable).
    """
    import pandas as pd

    N = meta_data.shape[0]
    if hasattr(data_mat, "values"):
        data_mat = data_mat.values
    data_mat = np.asarray(data_mat)
    if data_mat.shape[1] != N:                      # harmony.py:117-118
        data_mat = data_mat.T
    assert data_mat.shape[1] == N, \
        "data_mat and meta_data do not have the same number of cells"

    if nclust is None:                              # harmony.py:123-124
        nclust = int(min(round(N / 30.0), 100))
    if isinstance(sigma, float) and nclust > 1:     # harmony.py:126-127
        sigma = np.repeat(sigma, nclust)
    if isinstance(vars_use, str):
        vars_use = [vars_use]

    phi = pd.get_dummies(meta_data[vars_use]).to_numpy().T.astype(F32)   # :133
    phi_n = meta_data[vars_use].describe().loc["unique"].to_numpy().astype(int)  # :134

    if theta is None:                               # harmony.py:137-144
        theta = np.repeat([2] * len(phi_n), phi_n).astype(F32)
    elif isinstance(theta, (float, int)):
        theta = np.repeat([theta] * len(phi_n), phi_n).astype(F32)
    elif len(theta) == len(phi_n):
        theta = np.repeat([theta], phi_n).astype(F32)
    else:
        theta = np.asarray(theta, dtype=F32)
    assert len(theta) == np.sum(phi_n), "each batch variable must have a theta"

    lambda_estimation = False                       # harmony.py:150-166
    if lamb is None:
        lamb = np.insert(np.repeat([1] * len(phi_n), phi_n).astype(F32), 0, 0).astype(F32)
    elif np.isscalar(lamb) and lamb == -1:
        lambda_estimation = True
        lamb = np.zeros(1, dtype=F32)
    elif isinstance(lamb, (float, int)):
        lamb = np.insert(np.repeat([lamb] * len(phi_n), phi_n).astype(F32), 0, 0).astype(F32)
    elif len(lamb) == len(phi_n):
        lamb = np.insert(np.repeat([lamb], phi_n).astype(F32), 0, 0).astype(F32)
    else:
        lamb = np.asarray(lamb, dtype=F32)
        if len(lamb) == np.sum(phi_n):
            lamb = np.insert(lamb, 0, 0).astype(F32)

    N_b = phi.sum(axis=1)                           # harmony.py:169-170
    Pr_b = (N_b / N).astype(F32)
    if tau > 0:                                     # harmony.py:172-173
        theta = theta * (1 - np.exp(-(N_b / (nclust * tau)) ** 2))

    return dict(Z=np.asarray(data_mat, dtype=F32), phi=phi, Pr_b=Pr_b,
                sigma=np.asarray(sigma, dtype=F32), theta=np.asarray(theta, dtype=F32),
                lamb=lamb, lambda_estimation=lambda_estimation, K=int(nclust))


# --------------------------------------------------------------------------
# helpers: harmony.py:572-591
# --------------------------------------------------------------------------
def _x_log_x(x):
    """x*log(x) with non-finite entries zeroed (harmony.py:572-576)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = x * np.log(x)
    out[~np.isfinite(out)] = 0
    return out


def _col_pow(A, expo):
    """A[:, c] ** expo[c] per column (harmony.py:579-584)."""
    out = np.empty_like(A)
    for c in range(A.shape[1]):
        out[:, c] = np.power(A[:, c], expo[c])
    return out


def _col_unit(M):
    """Divide each column by its Euclidean length (harmony.py:238,377,444,569)."""
    return M / np.sqrt(np.sum(M * M, axis=0, dtype=F32), dtype=F32)


def kmeans_centroids(Z_cos, K, random_state):
    """The reference's host-side initialisation call (harmony.py:369-373)."""
    from sklearn.cluster import KMeans
    km = KMeans(n_clusters=K, init="k-means++", n_init=1, max_iter=25,
                random_state=random_state)
    km.fit(np.ascontiguousarray(Z_cos.T))
    return np.asarray(km.cluster_centers_.T, dtype=F32)     # d x K


class OracleHarmony:
    """State + methods of ``class Harmony`` (harmony.py:218-569), NumPy fp32.

    Unlike the reference the constructor does not run the algorithm by itself
    unless ``run=True``; tests drive the steps one at a time.  ``Y0`` (d x K)
    bypasses the sklearn call so that a test can feed identical centroids to
    oracle, reference and engine.  ``perm_source`` may be a callable
    ``N -> int64 permutation`` replacing ``torch.randperm`` (harmony.py:471).
    """

    def __init__(self, Z, Phi, Pr_b, sigma, theta, lamb, alpha=0.2,
                 lambda_estimation=False, max_iter_harmony=10, max_iter_kmeans=20,
                 epsilon_kmeans=1e-5, epsilon_harmony=1e-4, K=None, block_size=0.05,
                 random_state=0, Y0=None, perm_source=None, run=True, hooks=None,
                 forced_rounds=None, ridge_dtype=np.float32):
        # ridge_dtype=float64 is NOT the reference's arithmetic: it evaluates the same ridge
        # equations (harmony.py:547-566) in double.  The reference's fp32 inverse amplifies
        # rounding by cond(cov) ~ cluster mass / lambda: measured with the reference itself at
        # 69k cells, K=30: 9e-4 relative change of Z_corr between 1 and 8 host threads.  Tests
        # at that scale therefore check the engine against this better-conditioned evaluation.
        self.ridge_dtype = np.dtype(ridge_dtype)
        self.Z_orig = np.array(Z, dtype=F32)                     # d x N  (:235)
        self.Z_corr = self.Z_orig.copy()                         # (:234)
        self.Z_cos = _col_unit(self.Z_orig)                      # (:238)
        self.Phi = np.array(Phi, dtype=F32)                      # B x N  (:241)
        self.Pr_b = np.array(Pr_b, dtype=F32)
        self.d, self.N = self.Z_orig.shape
        self.B = self.Phi.shape[0]
        self.batch_index = [np.nonzero(self.Phi[b] > 0)[0] for b in range(self.B)]  # :249-252
        self.Phi_moe = np.concatenate([np.ones((1, self.N), F32), self.Phi], 0)     # :255-256
        self.window_size = 3
        self.epsilon_kmeans = epsilon_kmeans
        self.epsilon_harmony = epsilon_harmony
        self.lamb = np.array(lamb, dtype=F32)
        self.alpha = alpha
        self.lambda_estimation = lambda_estimation
        self.sigma = np.array(sigma, dtype=F32)
        self.theta = np.array(theta, dtype=F32)
        self.block_size = block_size
        self.K = int(K)
        self.max_iter_harmony = max_iter_harmony
        self.max_iter_kmeans = max_iter_kmeans
        self.objective_harmony = []
        self.objective_kmeans = []
        self.objective_kmeans_dist = []
        self.objective_kmeans_entropy = []
        self.objective_kmeans_cross = []
        self.kmeans_rounds = []
        self.hooks = hooks or {}
        self._perm_source = perm_source
        # test aid: replay a recorded round schedule instead of thresholding the
        # objective (harmony.py:455-458 decides on margins of a few fp32 ulps)
        self._forced_rounds = list(forced_rounds) if forced_rounds is not None else None
        # buffers of allocate_buffers (:357-364) that outlive a step
        self.R = np.zeros((self.K, self.N), F32)
        self.dist = np.zeros((self.K, self.N), F32)
        self.O = np.zeros((self.K, self.B), F32)
        self.E = np.zeros((self.K, self.B), F32)
        self.Y = np.zeros((self.d, self.K), F32)
        if run:
            self.init_cluster(random_state, Y0)
            self.harmonize(self.max_iter_harmony)

    # ---- hooks -----------------------------------------------------------
    def _emit(self, name):
        fn = self.hooks.get(name)
        if fn is not None:
            fn(self)

    def _randperm(self):
        if self._perm_source is not None:
            return np.asarray(self._perm_source(self.N), dtype=np.int64)
        import torch                                         # harmony.py:471, CPU generator
        return torch.randperm(self.N).numpy()

    # ---- harmony.py:366-392 ---------------------------------------------
    def init_cluster(self, random_state, Y0=None):
        if Y0 is None:
            Y0 = kmeans_centroids(self.Z_cos, self.K, random_state)
        self.Y0 = np.array(Y0, dtype=F32)
        self.Y = _col_unit(self.Y0)                                    # :377
        self.dist = (F32(2) * (F32(1) - self.Y.T @ self.Z_cos)).astype(F32)   # :380
        R = np.exp(-self.dist / self.sigma[:, None])                   # :383-384
        self.R = (R / R.sum(axis=0, dtype=F32)).astype(F32)            # :385
        self.E = np.outer(self.R.sum(axis=1, dtype=F32), self.Pr_b).astype(F32)   # :388
        self.O = (self.R @ self.Phi.T).astype(F32)                     # :389
        self.compute_objective()                                       # :391
        self.objective_harmony.append(self.objective_kmeans[-1])       # :392
        self._emit("init_cluster")

    # ---- harmony.py:394-417 ---------------------------------------------
    def compute_objective(self):
        norm_const = 2000.0 / self.N
        kmeans_error = float(np.sum(self.R * self.dist, dtype=F32))                     # :399
        entropy = float(np.sum(_x_log_x(self.R) * self.sigma[:, None], dtype=F32))      # :402
        R_sigma = self.R * self.sigma[:, None]                                          # :405
        O_c = np.maximum(self.O, F32(1e-8))                                             # :407
        E_c = np.maximum(self.E, F32(1e-8))                                             # :408
        ratio = (O_c + E_c) / E_c                                                       # :409
        theta_log = self.theta[None, :] * np.log(ratio)                                 # :410
        cross = float(np.sum(R_sigma * (theta_log @ self.Phi), dtype=F32))              # :411
        self.objective_kmeans.append((kmeans_error + entropy + cross) * norm_const)     # :414
        self.objective_kmeans_dist.append(kmeans_error * norm_const)
        self.objective_kmeans_entropy.append(entropy * norm_const)
        self.objective_kmeans_cross.append(cross * norm_const)

    # ---- harmony.py:419-435 ---------------------------------------------
    def harmonize(self, iter_harmony=10):
        converged = False
        for _ in range(1, iter_harmony + 1):
            self.cluster()
            self.moe_correct_ridge()
            converged = self.check_convergence(1)
            if converged:
                break
        return converged

    # ---- harmony.py:437-462 ---------------------------------------------
    def cluster(self):
        self.dist = (F32(2) * (F32(1) - self.Y.T @ self.Z_cos)).astype(F32)       # :438
        rounds = 0
        forced = self._forced_rounds.pop(0) if self._forced_rounds else None
        for i in range(self.max_iter_kmeans if forced is None else forced):
            self.Y = _col_unit((self.Z_cos @ self.R.T).astype(F32))               # :443-444
            self.dist = (F32(2) * (F32(1) - self.Y.T @ self.Z_cos)).astype(F32)   # :447
            self.update_R()                                                       # :450
            self.compute_objective()                                              # :453
            self._emit("round")
            if forced is None and i > self.window_size and self.check_convergence(0):   # :455-458
                rounds = i + 1
                break
            rounds = i + 1
        self.kmeans_rounds.append(rounds)                                         # :461
        self.objective_harmony.append(self.objective_kmeans[-1])                  # :462

    # ---- harmony.py:464-513 ---------------------------------------------
    def update_R(self):
        scale = np.exp(-self.dist / self.sigma[:, None])                 # :466-467
        scale = (scale / scale.sum(axis=0, dtype=F32)).astype(F32)       # :468
        order = self._randperm()                                         # :471
        self.last_order = order
        n_blocks = int(np.ceil(1.0 / self.block_size))                   # :474
        per_block = int(self.N * self.block_size)                        # :475
        # NumPy's fancy indexing hands back column-major copies; torch's gathers are row-major, and
        # only then are the fp32 row sums below pairwise like torch's (a strided fp32 sum runs
        # sequentially, drops R's many tiny entries and biases E by ~3e-5 per round at 150k cells)
        R_p = np.ascontiguousarray(self.R[:, order])                     # :478
        scale_p = np.ascontiguousarray(scale[:, order])                  # :479
        Phi_p = np.ascontiguousarray(self.Phi[:, order])                 # :480
        for blk in range(n_blocks):
            lo = blk * per_block                                         # :483
            hi = self.N if blk == n_blocks - 1 else (blk + 1) * per_block  # :484
            R_b = R_p[:, lo:hi]
            Phi_b = Phi_p[:, lo:hi]
            # take the block out of the statistics (:491-492)
            self.E = self.E - np.outer(R_b.sum(axis=1, dtype=F32), self.Pr_b).astype(F32)
            self.O = self.O - (R_b @ Phi_b.T).astype(F32)
            # diversity-penalised reassignment (:495-503)
            OE = np.maximum(self.O + self.E, F32(1e-8))
            ratio = np.clip(self.E / OE, F32(1e-8), F32(1.0))
            ratio_pow = _col_pow(ratio, self.theta)
            R_new = scale_p[:, lo:hi] * (ratio_pow @ Phi_b)
            col = np.maximum(R_new.sum(axis=0, dtype=F32), F32(1e-8))
            R_new = (R_new / col).astype(F32)
            # put the block back (:506-507)
            self.E = self.E + np.outer(R_new.sum(axis=1, dtype=F32), self.Pr_b).astype(F32)
            self.O = self.O + (R_new @ Phi_b.T).astype(F32)
            R_p[:, lo:hi] = R_new                                        # :509
        inverse = np.argsort(order, kind="stable")                       # :512
        self.R = np.ascontiguousarray(R_p[:, inverse])                   # :513
        self._emit("update_R")

    # ---- harmony.py:515-533 ---------------------------------------------
    def check_convergence(self, i_type):
        if i_type == 0:
            if len(self.objective_kmeans) <= self.window_size + 1:
                return False
            w = self.window_size
            obj_old = sum(self.objective_kmeans[-w - 1:-1])
            obj_new = sum(self.objective_kmeans[-w:])
            return abs(obj_old - obj_new) / abs(obj_old) < self.epsilon_kmeans
        if i_type == 1:
            if len(self.objective_harmony) < 2:
                return False
            obj_old = self.objective_harmony[-2]
            obj_new = self.objective_harmony[-1]
            return (obj_old - obj_new) / abs(obj_old) < self.epsilon_harmony
        return True

    # ---- harmony.py:535-569 ---------------------------------------------
    def moe_correct_ridge(self):
        T = self.ridge_dtype.type
        Z_orig = self.Z_orig.astype(T)
        Phi_moe = self.Phi_moe.astype(T)
        Z_corr = Z_orig.copy()                                           # :537
        self.W_all = np.zeros((self.K, self.B + 1, self.d), F32)
        for k in range(self.K):
            if self.lambda_estimation:                                   # :541-544, 587-591
                lam = np.zeros(self.B + 1, F32)
                lam[1:] = self.E[k, :] * F32(self.alpha)
            else:
                lam = self.lamb
            Rk = self.R[k, :].astype(T)
            Phi_Rk = Phi_moe * Rk                                        # :547
            cov = (Phi_Rk @ Phi_moe.T + np.diag(lam.astype(T))).astype(T)   # :550
            inv_cov = np.linalg.inv(cov).astype(T)                       # :553
            Z_tmp = Z_orig * Rk                                          # :556
            # The row sums over N are accumulated in float64 and rounded once to the working type: torch's sum kernels
            # are that accurate, NumPy's float32 row sums are not -- at BASELINE configs[4]'s shape (signed PCs, heavy
            # cancellation) they alone moved Z_corr by 1.2e-4 away from the reference, which sits 3.6e-6 from its own
            # float64 evaluation there (tests/golden/ridge_conditioning.json, tests/test_large_golden.py).
            W = inv_cov[:, 0:1] @ Z_tmp.sum(axis=1, dtype=np.float64, keepdims=True).astype(T).T   # :559
            for b in range(self.B):                                      # :561-563
                cols = np.ascontiguousarray(Z_tmp[:, self.batch_index[b]])
                part = cols.sum(axis=1, dtype=np.float64, keepdims=True).astype(T)
                W = W + inv_cov[:, b + 1:b + 2] @ part.T
            W[0, :] = 0                                                  # :565
            self.W_all[k] = W
            Z_corr = (Z_corr - W.T @ Phi_Rk).astype(T)                   # :566
        self.Z_corr = Z_corr.astype(F32)
        self.Z_cos = _col_unit(self.Z_corr)                              # :569
        self._emit("ridge")

    # ---- read-back in the reference's public orientation (:288-355) ------
    def result(self):
        return self.Z_corr.T


def oracle_run_harmony(data_mat, meta_data, vars_use, theta=None, lamb=None, sigma=0.1,
                       nclust=None, tau=0, block_size=0.05, max_iter_harmony=10,
                       max_iter_kmeans=20, epsilon_cluster=1e-5, epsilon_harmony=1e-4,
                       alpha=0.2, random_state=0, Y0=None, hooks=None, forced_rounds=None,
                       ridge_dtype=np.float32):
    """``run_harmony`` (harmony.py:49-215) on the oracle; seeds like :199-200."""
    import torch
    p = prepare_inputs(data_mat, meta_data, vars_use, theta, lamb, sigma, nclust, tau)
    np.random.seed(random_state)
    torch.manual_seed(random_state)
    return OracleHarmony(p["Z"], p["phi"], p["Pr_b"], p["sigma"], p["theta"], p["lamb"],
                         alpha, p["lambda_estimation"], max_iter_harmony, max_iter_kmeans,
                         epsilon_cluster, epsilon_harmony, p["K"], block_size,
                         random_state, Y0=Y0, hooks=hooks, forced_rounds=forced_rounds,
                         ridge_dtype=ridge_dtype)
