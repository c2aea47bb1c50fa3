"""Cell-sharded restatement of the oracle (CPU, NumPy).  TEST INFRASTRUCTURE ONLY.

States, on the CPU, the claim the sharded engine is built on: if every rank holds
a slice of the cells and the ranks sum ONLY the small cross-cell tables

    init      R.sum(1) (K), R Phi^T (K x B)                     harmony.py:388-389
    round     Z_cos R^T (d x K)                                  harmony.py:443
    block     R_blk.sum(1), R_blk Phi_blk^T, before and after    harmony.py:491-492, 506-507
    objective the three scalar sums                              harmony.py:399-411
    ridge     Phi_Rk Phi_moe^T ((B+1)^2), Z_tmp sums (d, B x d)  harmony.py:550, 559-563

then every rank reproduces the unsharded run on its own cells.  ``ShardedOracle`` is
``OracleHarmony`` with exactly those sums routed through ``reduce`` (an in-place sum over the
ranks, e.g. ``torch.distributed.all_reduce`` on gloo); blocks are cut from ONE permutation of all
cells of the job (harmony.py:471-484), each rank keeping its own members.  The tests run it with
world_size 2 and compare with the unsharded oracle.
"""
from __future__ import annotations

import numpy as np

from .harmony_oracle import F32, OracleHarmony, _col_pow, _col_unit, _x_log_x


class ShardedOracle(OracleHarmony):
    def __init__(self, Z, Phi, Pr_b, sigma, theta, lamb, *, reduce, offset, n_global, **kw):
        self._reduce = reduce
        self._offset = int(offset)
        self.N_global = int(n_global)
        kw["run"] = False
        super().__init__(Z, Phi, Pr_b, sigma, theta, lamb, **kw)

    def _sum(self, x):
        """Sum a small table over the ranks (fp64 on the wire, like the engine), back to fp32."""
        buf = np.ascontiguousarray(x, dtype=np.float64)
        self._reduce(buf)
        return buf.astype(F32)

    # harmony.py:376-392
    def init_cluster(self, random_state, Y0=None):
        assert Y0 is not None, "the sharded oracle takes the centroids of the unsharded fit"
        self.Y0 = np.array(Y0, dtype=F32)
        self.Y = _col_unit(self.Y0)
        self.dist = (F32(2) * (F32(1) - self.Y.T @ self.Z_cos)).astype(F32)
        R = np.exp(-self.dist / self.sigma[:, None])
        self.R = (R / R.sum(axis=0, dtype=F32)).astype(F32)
        self.E = np.outer(self._sum(self.R.sum(axis=1, dtype=F32)), self.Pr_b).astype(F32)
        self.O = self._sum(self.R @ self.Phi.T)
        self.compute_objective()
        self.objective_harmony.append(self.objective_kmeans[-1])
        self._emit("init_cluster")

    # harmony.py:394-417
    def compute_objective(self):
        norm_const = 2000.0 / self.N_global
        R_sigma = self.R * self.sigma[:, None]
        O_c = np.maximum(self.O, F32(1e-8))
        E_c = np.maximum(self.E, F32(1e-8))
        theta_log = self.theta[None, :] * np.log((O_c + E_c) / E_c)
        parts = np.array([np.sum(self.R * self.dist, dtype=np.float64),
                          np.sum(_x_log_x(self.R) * self.sigma[:, None], dtype=np.float64),
                          np.sum(R_sigma * (theta_log @ self.Phi), dtype=np.float64)])
        self._reduce(parts)
        kmeans_error, entropy, cross = (float(F32(v)) for v in parts)
        self.objective_kmeans.append((kmeans_error + entropy + cross) * norm_const)
        self.objective_kmeans_dist.append(kmeans_error * norm_const)
        self.objective_kmeans_entropy.append(entropy * norm_const)
        self.objective_kmeans_cross.append(cross * norm_const)

    # harmony.py:437-462
    def cluster(self):
        rounds = 0
        forced = self._forced_rounds.pop(0) if self._forced_rounds else None
        for i in range(self.max_iter_kmeans if forced is None else forced):
            self.Y = _col_unit(self._sum(self.Z_cos @ self.R.T))
            self.dist = (F32(2) * (F32(1) - self.Y.T @ self.Z_cos)).astype(F32)
            self.update_R()
            self.compute_objective()
            self._emit("round")
            if forced is None and i > self.window_size and self.check_convergence(0):
                rounds = i + 1
                break
            rounds = i + 1
        self.kmeans_rounds.append(rounds)
        self.objective_harmony.append(self.objective_kmeans[-1])

    def _randperm(self):
        if self._perm_source is not None:
            return np.asarray(self._perm_source(self.N_global), dtype=np.int64)
        import torch
        return torch.randperm(self.N_global).numpy()

    # harmony.py:464-513
    def update_R(self):
        scale = np.exp(-self.dist / self.sigma[:, None])
        scale = (scale / scale.sum(axis=0, dtype=F32)).astype(F32)
        order = self._randperm()                                   # one order of ALL cells, same on every rank
        n_blocks = int(np.ceil(1.0 / self.block_size))
        per_block = int(self.N_global * self.block_size)
        for blk in range(n_blocks):
            lo = blk * per_block
            hi = self.N_global if blk == n_blocks - 1 else (blk + 1) * per_block
            members = order[lo:hi]
            mine = members[(members >= self._offset) & (members < self._offset + self.N)] - self._offset
            # row-major gathers: fp32 row sums are then pairwise like torch's (harmony_oracle.update_R)
            R_b, Phi_b = np.ascontiguousarray(self.R[:, mine]), np.ascontiguousarray(self.Phi[:, mine])
            self.E = self.E - np.outer(self._sum(R_b.sum(axis=1, dtype=F32)), self.Pr_b).astype(F32)
            self.O = self.O - self._sum(R_b @ Phi_b.T)
            OE = np.maximum(self.O + self.E, F32(1e-8))
            ratio = np.clip(self.E / OE, F32(1e-8), F32(1.0))
            ratio_pow = _col_pow(ratio, self.theta)
            R_new = np.ascontiguousarray(scale[:, mine]) * (ratio_pow @ Phi_b)
            col = np.maximum(R_new.sum(axis=0, dtype=F32), F32(1e-8))
            R_new = (R_new / col).astype(F32)
            self.E = self.E + np.outer(self._sum(R_new.sum(axis=1, dtype=F32)), self.Pr_b).astype(F32)
            self.O = self.O + self._sum(R_new @ Phi_b.T)
            self.R[:, mine] = R_new
        self._emit("update_R")

    # harmony.py:535-569
    def moe_correct_ridge(self):
        T = self.ridge_dtype.type
        Z_orig = self.Z_orig.astype(T)
        Phi_moe = self.Phi_moe.astype(T)
        Z_corr = Z_orig.copy()
        for k in range(self.K):
            if self.lambda_estimation:
                lam = np.zeros(self.B + 1, F32)
                lam[1:] = self.E[k, :] * F32(self.alpha)
            else:
                lam = self.lamb
            Rk = self.R[k, :].astype(T)
            Phi_Rk = Phi_moe * Rk
            cov = (self._sum(Phi_Rk @ Phi_moe.T).astype(T) + np.diag(lam.astype(T))).astype(T)
            inv_cov = np.linalg.inv(cov).astype(T)
            Z_tmp = Z_orig * Rk
            sums = np.stack([Z_tmp.sum(axis=1, dtype=T)] +
                            [Z_tmp[:, self.batch_index[b]].sum(axis=1, dtype=T) for b in range(self.B)])
            sums = self._sum(sums).astype(T)                               # (B+1) x d
            W = inv_cov @ sums
            W[0, :] = 0
            Z_corr = (Z_corr - W.T @ Phi_Rk).astype(T)
        self.Z_corr = Z_corr.astype(F32)
        self.Z_cos = _col_unit(self.Z_corr)
        self._emit("ridge")
