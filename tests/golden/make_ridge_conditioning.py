#!/usr/bin/env python3
"""Evidence for DESIGN.md §5 "the reference's fp32 ridge is ill-conditioned at large N".

Runs ONLY in the build container (imports /root/reference, harmonypy v0.2.0, device='cpu').
BASELINE configs[1] shape (69k cells x 50 PCs, 4 batches, K=30), identical initial centroids (the
sklearn fit is replaced by a fixed Y0), 5 k-means rounds (epsilon_cluster=0), ONE ridge correction.
The reference is run with 1 and with 8 torch threads, and once with harmony.py:553's inverse
evaluated in float64 (with 1 and with 8 threads as well).  Writes tests/golden/ridge_conditioning.json:

  R_relF_1_vs_8_threads        -- what reaches the ridge step differs by fp32 summation noise only
  Zcorr_relF_1_vs_8_threads    -- ... and the fp32 ridge turns that into this
  Zcorr_relF_f32_vs_f64_inverse
  Zcorr_relF_f64_inverse_1_vs_8_threads -- self-noise of the float64-inverse variant (as large: the noise is the fp32
                                  SUMMATION of cov and of the right-hand sides, amplified by cond(cov), not the inverse)
  Zcorr_relF_f64_ridge_1_vs_8_threads   -- self-noise of (iii): the reference's own moe_correct_ridge run on float64 copies
  cond_cov_median / max        -- condition number of cov (harmony.py:550) over the K clusters

and, per shape, tests/golden/large_<name>.npz -- outputs of the REFERENCE ITSELF that pin the large-N parity tests
(tests/test_large_golden.py): Y0, the four objective histories, R column sums, a 2k-row sample of R, and the same
rows of Z_corr from (i) the plain reference, (ii) the reference with harmony.py:553's inverse taken in float64 and
(iii) the reference's own moe_correct_ridge run on float64 copies of its tensors (everything else untouched), on 1 thread.  Shapes: BASELINE configs[1] (69k x 50, K=30), configs[2]'s shape at
150k cells (K=100, 8 batches) and configs[4]'s shape at 40k cells (200 PCs, K=200, 32 batches).

    python tests/golden/make_ridge_conditioning.py
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import logging  # noqa: E402
import torch  # noqa: E402
import harmonypy as hm  # noqa: E402
import harmonypy.harmony as hh  # noqa: E402
from bench import quick_centroids, synthetic_dataset  # noqa: E402

logging.getLogger("harmonypy").setLevel(logging.WARNING)
_state = {"Y0": None, "conds": [], "last": None}


class _FixedKMeans:
    """Stands in for sklearn.KMeans at harmony.py:370-372: hands back the prepared centroids."""
    def __init__(self, *a, **k):
        pass

    def fit(self, X):
        self.cluster_centers_ = _state["Y0"].T.astype(np.float64)
        return self


hh.KMeans = _FixedKMeans
_inv = torch.linalg.inv


_ridge = hh.Harmony.moe_correct_ridge


def _ridge_in_float64(self):
    """The reference's OWN moe_correct_ridge (harmony.py:535-569), unchanged, run on float64 copies of the tensors it
    reads: variant (iii), the one evaluation of those equations that does not depend on the summation order."""
    names = ["_Z_orig", "_R", "_Phi_moe", "_lamb", "_E"]
    saved = {n: getattr(self, n) for n in names}
    for n in names:
        setattr(self, n, saved[n].double())
    try:
        _ridge(self)
    finally:
        for n in names:
            setattr(self, n, saved[n])
    self._Z_corr = self._Z_corr.float()
    self._Z_cos = self._Z_cos.float()


def run(threads, inv64=False, ridge64=False, N=69_000, d=50, B=4, K=30, seed=0):
    torch.set_num_threads(threads)
    hh.Harmony.moe_correct_ridge = _ridge_in_float64 if ridge64 else _ridge
    Z, meta = synthetic_dataset(N, d, B, K, seed=seed)
    if _state["Y0"] is None or _state["Y0"].shape != (d, K):
        _state["Y0"] = quick_centroids(Z, K, seed=seed, sample=20_000 if seed else 50_000)

    def inv(x):
        _state["conds"].append(float(np.linalg.cond(x.double().numpy())))
        return _inv(x.double()).float() if inv64 else _inv(x)
    torch.linalg.inv = inv
    try:
        ho = hm.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=1, max_iter_kmeans=5, epsilon_cluster=0.0,
                            epsilon_harmony=-1e30, verbose=False, random_state=0, device="cpu")
    finally:
        torch.linalg.inv = _inv
        hh.Harmony.moe_correct_ridge = _ridge
    _state["last"] = ho
    return ho.R.copy(), ho.Z_corr.copy()


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


SAMPLE_ROWS = 2000


def case(label, name=None, **shape):
    _state["conds"].clear()
    R1, Z1 = run(1, **shape)
    ho1 = _state["last"]
    conds = list(_state["conds"])
    R8, Z8 = run(8, **shape)
    _, Z64 = run(1, inv64=True, **shape)
    _, Z64_8 = run(8, inv64=True, **shape)
    _, Zr64 = run(1, ridge64=True, **shape)
    _, Zr64_8 = run(8, ridge64=True, **shape)
    if name:
        N = R1.shape[0]
        n_rows = SAMPLE_ROWS if R1.shape[1] * Z1.shape[1] <= 5000 else 800      # keeps every file near 2 MB
        rows = np.linspace(0, N - 1, min(n_rows, N)).astype(np.int64)
        np.savez_compressed(
            os.path.join(HERE, f"large_{name}.npz"),
            shape=np.array([shape.get("N", 69_000), shape.get("d", 50), shape.get("B", 4), shape.get("K", 30),
                            shape.get("seed", 0)], dtype=np.int64),
            Y0=_state["Y0"].astype(np.float32), rows=rows,
            R_rows=R1[rows].astype(np.float32), R_colsum=R1.astype(np.float64).sum(axis=0),
            O=ho1.O.astype(np.float32), E=ho1.E.astype(np.float32),
            objective_kmeans=np.asarray(ho1.objective_kmeans, dtype=np.float64),
            objective_kmeans_dist=np.asarray(ho1.objective_kmeans_dist, dtype=np.float64),
            objective_kmeans_entropy=np.asarray(ho1.objective_kmeans_entropy, dtype=np.float64),
            objective_kmeans_cross=np.asarray(ho1.objective_kmeans_cross, dtype=np.float64),
            objective_harmony=np.asarray(ho1.objective_harmony, dtype=np.float64),
            kmeans_rounds=np.asarray(ho1.kmeans_rounds, dtype=np.int64),
            Zcorr_rows_plain=Z1[rows].astype(np.float32), Zcorr_rows_inv64=Z64[rows].astype(np.float32),
            Zcorr_rows_ridge64=Zr64[rows].astype(np.float32),
            Zcorr_absmax=np.float64(np.abs(Zr64).max()),
            Zcorr_norm_ridge64=np.float64(np.linalg.norm(Zr64.astype(np.float64))),
            plain_relF_1_vs_8_threads=np.float64(rel(Z8, Z1)),
            Zcorr_norm_plain=np.float64(np.linalg.norm(Z1.astype(np.float64))),
            Zcorr_norm_inv64=np.float64(np.linalg.norm(Z64.astype(np.float64))))
    return {
        "Zcorr_relF_f64_ridge_1_vs_8_threads": rel(Zr64_8, Zr64),
        "Zcorr_maxabs_over_max_f64_ridge_1_vs_8_threads": float(np.abs(Zr64_8 - Zr64).max() / np.abs(Zr64).max()),
        "Zcorr_relF_plain_vs_f64_ridge": rel(Z1, Zr64),
        "Zcorr_relF_f64_inverse_1_vs_8_threads": rel(Z64_8, Z64),
        "Zcorr_maxabs_over_max_f64_inverse_1_vs_8_threads": float(np.abs(Z64_8 - Z64).max() / np.abs(Z64).max()),
        "shape": label + "; 5 rounds + 1 ridge, same Y0, same randperm stream",
        "R_relF_1_vs_8_threads": rel(R8, R1),
        "Zcorr_relF_1_vs_8_threads": rel(Z8, Z1),
        "Zcorr_maxabs_over_max_1_vs_8_threads": float(np.abs(Z8 - Z1).max() / np.abs(Z1).max()),
        "Zcorr_relF_f32_vs_f64_inverse": rel(Z1, Z64),
        "Zcorr_maxabs_over_max_f32_vs_f64_inverse": float(np.abs(Z1 - Z64).max() / np.abs(Z64).max()),
        "cond_cov_median": float(np.median(conds)), "cond_cov_max": float(np.max(conds)),
    }


def main():
    out = {
        "reference": "harmonypy v0.2.0 at /root/reference, device='cpu', torch " + torch.__version__,
        "configs_1": case("69000 cells x 50 PCs, 4 batches, K=30 (BASELINE configs[1])", name="c2"),
        "configs_2_shape": case("150000 cells x 50 PCs, 8 batches, K=100 (BASELINE configs[2] shape)", name="c3shape",
                                N=150_000, d=50, B=8, K=100, seed=3),
        # shape of tests/test_parity_gpu.py::test_bench_path_parity_c5_shape: lamb[0] = 0 (harmony.py:150-152) leaves
        # cov[0,0] = the cluster's mass, and K=200 clusters over 100 cell types leave many clusters nearly empty
        "configs_4_shape": case("40000 cells x 200 PCs, 32 batches, K=200 (BASELINE configs[4] shape)", name="c5shape",
                                N=40_000, d=200, B=32, K=200, seed=3),
    }
    with open(os.path.join(HERE, "ridge_conditioning.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
